"""Host loop of corpus / question encoding: the counterpart of predict() and the np.save in
reference scripts/encode_corpus.py:116-151.

    embeds = predict(model, eval_dataloader)          # [N, D] float32, dataset order
    save_embeddings(path, embeds)                     # np.save, row i <-> id2doc[str(i)]

Unlike the reference (its `try/except ... continue`, :138-143) a failing batch raises: dropping a
batch would silently shift every later row against id2doc.json
(retrieval/data/encode_datasets.py:145-152).
"""
import os

import numpy as np
import torch

from .encoder import move_to_cuda


def predict(model, eval_dataloader, to_cpu=True):
    model.eval()
    embed_array = []
    for batch in eval_dataloader:
        batch = move_to_cuda(batch)
        embed = model(batch)["embed"]
        embed_array.append(embed.cpu() if to_cpu else embed)
    if not embed_array:
        raise ValueError("predict: the dataloader produced no batches")
    return torch.cat(embed_array, dim=0)


def save_embeddings(path, embeds):
    """np.save(args.embed_save_path, embeds.cpu().numpy()) — scripts/encode_corpus.py:119-123."""
    d = os.path.dirname(path)
    if d and not os.path.exists(d):
        os.makedirs(d)
    np.save(path, embeds.detach().cpu().numpy().astype(np.float32, copy=False))


def collate_tokens(values, pad_idx):
    """retrieval/data/data_utils.py:5-31 (1-d case used by em_collate_bert): right-pad to the batch max."""
    values = [v.view(-1) for v in values]
    size = max(v.size(0) for v in values)
    res = values[0].new_full((len(values), size), pad_idx)
    for i, v in enumerate(values):
        res[i, : v.numel()] = v
    return res


def em_collate_bert(samples, pad_id=0):
    """retrieval/data/encode_datasets.py:393-417: list of dataset items ({'tb': tokenizer features})
    -> batch dict {input_ids, input_mask[, part2_mask, part3_mask, input_type_ids]}."""
    if len(samples) == 0:
        return {}
    batch = {
        "input_ids": collate_tokens([s["tb"]["input_ids"].view(-1) for s in samples], pad_id),
        "input_mask": collate_tokens([s["tb"]["attention_mask"].view(-1) for s in samples], 0),
    }
    for src, dst in (("part2_mask", "part2_mask"), ("part3_mask", "part3_mask"), ("token_type_ids", "input_type_ids")):
        if src in samples[0]["tb"]:
            batch[dst] = collate_tokens([s["tb"][src].view(-1) for s in samples], 0)
    return batch
