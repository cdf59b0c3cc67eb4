"""kegg.py:252-327 label construction: the vectorised form against the position-by-position restatement, on rows with
planted markers (adjacent, nested, unterminated, at the row ends, none at all) and left padding."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bioreason_amd.collate import assistant_label_mask      # noqa: E402
from oracle.collate_oracle import assistant_labels_loop     # noqa: E402

A, E, PAD = [7, 8, 9], [5], 0


def _rows(seed, B=6, L=48):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(10, 60, (B, L), generator=g)
    for b in range(B):
        for _ in range(int(torch.randint(0, 4, (1,), generator=g))):
            p = int(torch.randint(0, L - 3, (1,), generator=g))
            ids[b, p:p + 3] = torch.tensor(A)
        for _ in range(int(torch.randint(0, 4, (1,), generator=g))):
            ids[b, int(torch.randint(0, L, (1,), generator=g))] = E[0]
        npad = int(torch.randint(0, 6, (1,), generator=g))
        ids[b, :npad] = PAD                                   # padding_side="left" (kegg.py:246)
    return ids


@pytest.mark.parametrize("seed", range(12))
def test_label_mask_equals_reference_loop(seed):
    ids = _rows(seed)
    assert torch.equal(assistant_label_mask(ids, A, E, PAD), assistant_labels_loop(ids, A, E, PAD))


def test_label_mask_edge_rows():
    rows = [
        [7, 8, 9, 11, 12, 5, 13, 14],          # one closed section
        [11, 7, 8, 9, 12, 13, 14, 15],         # unterminated: runs to the row end
        [11, 12, 13, 14, 15, 7, 8, 9],         # marker at the very end: empty section
        [7, 8, 9, 5, 11, 7, 8, 9],             # end marker right after the start: empty; second start unterminated & empty
        [5, 5, 7, 8, 9, 11, 5, 12],            # end markers before any start are ignored
        [7, 8, 9, 7, 8, 9, 11, 5],             # two starts, one end: both sections close at it
        [0, 0, 7, 8, 9, 11, 12, 5],            # left padding
        [11, 12, 13, 14, 15, 16, 17, 18],      # no marker at all
    ]
    ids = torch.tensor(rows)
    got = assistant_label_mask(ids, A, E, PAD)
    assert torch.equal(got, assistant_labels_loop(ids, A, E, PAD))
    assert got[0].tolist() == [-100, -100, -100, 11, 12, -100, -100, -100]
    assert got[1].tolist() == [-100, -100, -100, -100, 12, 13, 14, 15]
    assert (got[2] == -100).all() and (got[7] == -100).all()
    # multi-token end marker and a marker longer than the row
    ids2 = torch.tensor([[7, 8, 9, 11, 5, 6, 12, 13]])
    assert torch.equal(assistant_label_mask(ids2, A, [5, 6], PAD), assistant_labels_loop(ids2, A, [5, 6], PAD))
    assert (assistant_label_mask(ids2[:, :2], A, E, PAD) == -100).all()
