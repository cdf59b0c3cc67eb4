"""TEST INFRASTRUCTURE ONLY — loop restatement of the label construction in the reference's `qwen_dna_collate_fn`
(bioreason/dataset/kegg.py:252-327), position by position as the reference does it, for checking
`bioreason_amd.collate.assistant_label_mask`.  Pinned: `tests/test_collate.py` runs the reference function itself when
/root/reference is importable (its tokenizer calls are replaced by the given marker ids)."""
import torch


def assistant_labels_loop(input_ids, assistant_start_token_ids, im_end_token_ids, pad_token_id):
    labels = torch.full_like(input_ids, -100)                                    # kegg.py:253
    a_len, e_len = len(assistant_start_token_ids), len(im_end_token_ids)
    a_t, e_t = torch.tensor(assistant_start_token_ids), torch.tensor(im_end_token_ids)
    for i in range(input_ids.shape[0]):                                          # kegg.py:279
        row = input_ids[i]
        L = row.size(0)
        starts = [p + a_len for p in range(L - a_len + 1) if torch.all(row[p:p + a_len] == a_t)]     # :285-292
        ends = [p for p in range(L - e_len + 1) if torch.all(row[p:p + e_len] == e_t)]               # :295-300
        sections = []
        for s in starts:                                                         # :303-313
            valid = [p for p in ends if p > s]
            if valid:
                e = min(valid)
                if s < e:
                    sections.append((s, e))
            else:
                sections.append((s, L))
        for s, e in sections:                                                    # :316-319
            if s < e and s < L:
                labels[i, s:min(e, L)] = row[s:min(e, L)]
    labels[input_ids == pad_token_id] = -100                                     # :322
    return labels
