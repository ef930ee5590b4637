cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for t in "" hold1 hold2 hold5; do
  if [ -z "$t" ]; then L=""; else L=$PWD/variants/libotter_$t.so; fi
  echo "== ${t:-hold3}: $(OTTER_LIB=$L timeout -k 10 150 python3 tools/bench_search_sweep.py --nq 33,64 --steps 30 2>/dev/null | cut -c1-76 | tr '\n' ' ')" || exit 1
done; done
