"""Generate tests/golden/data_golden.json by IMPORTING the reference's data_utils.py (DistributedBucketSampler,
TextMelMyOwnCollate) where it lies.  data_utils imports `librosa` (absent from this image) and the `text` package
(whose cleaners need unidecode / pyopenjtalk / jamo ..., absent too); neither is touched by the sampler or the collate,
so both names are registered in sys.modules as placeholders whose functions raise.  Nothing of the reference is copied.

    python tests/golden/make_data_golden.py
"""
import json
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("GLOWTTS_REFERENCE", "/root/reference")


def import_data_utils():
    def _never(*a, **k):
        raise RuntimeError("placeholder: not available in this image, not used by the sampler / collate")
    lib = types.ModuleType("librosa")
    lib.filters = types.ModuleType("librosa.filters"); lib.filters.mel = _never
    lib.util = types.ModuleType("librosa.util"); lib.util.pad_center = _never; lib.util.tiny = _never; lib.util.normalize = _never
    lib.stft = _never; lib.istft = _never
    txt = types.ModuleType("text"); txt.text_to_sequence = _never; txt.cleaned_text_to_sequence = _never
    txt.cmudict = types.ModuleType("text.cmudict")
    sym = types.ModuleType("text.symbols"); sym.symbols = []
    txt.symbols = sym
    for n, m in (("librosa", lib), ("librosa.filters", lib.filters), ("librosa.util", lib.util), ("text", txt),
                 ("text.symbols", sym), ("text.cmudict", txt.cmudict)):
        sys.modules.setdefault(n, m)
    sys.path.insert(0, REF)
    import data_utils
    return data_utils


class _Lengths:
    def __init__(self, lengths):
        self.lengths = lengths

    def __len__(self):
        return len(self.lengths)


def main():
    du = import_data_utils()
    out = {"sampler": [], "collate": []}
    g = torch.Generator().manual_seed(1234)
    cases = [(200, 8, [32, 300, 400, 500, 600, 700, 800, 900, 1000], 2, True),        # the reference's boundaries (train script)
             (57, 4, [32, 300, 400, 500, 600, 700, 800, 900, 1000], 3, True),
             (40, 4, [0, 100, 200, 300], 1, False),
             (33, 5, [10, 50, 60, 2000], 4, True)]                                    # an empty middle bucket is merged away
    for n, bs, bounds, world, shuffle in cases:
        lengths = torch.randint(20, 1100, (n,), generator=g).tolist()
        if bounds[1] == 50:
            lengths = [v if not (50 < v <= 60) else 61 for v in lengths]
        for rank in range(world):
            for epoch in (0, 3):
                s = du.DistributedBucketSampler(_Lengths(lengths), bs, list(bounds), num_replicas=world, rank=rank, shuffle=shuffle)
                s.set_epoch(epoch)
                batches = list(iter(s))
                out["sampler"].append(dict(lengths=lengths, batch_size=bs, boundaries=bounds, world=world, rank=rank, epoch=epoch,
                                           shuffle=shuffle, batches=batches, len=len(s), boundaries_after=list(s.boundaries)))
    col = du.TextMelMyOwnCollate(1)
    for n_frames in (1, 2):
        col = du.TextMelMyOwnCollate(n_frames)
        items = []
        for k in range(5):
            tl, ml = int(torch.randint(3, 12, (1,), generator=g)), int(torch.randint(5, 20, (1,), generator=g))
            items.append((torch.randint(1, 100, (tl,), generator=g), torch.randn(4, ml, generator=g), torch.randn(512, generator=g),
                          int(torch.randint(0, 5, (1,), generator=g)), torch.rand(3, generator=g), torch.rand(1, ml, generator=g),
                          torch.rand(1, ml, generator=g), int(torch.randint(0, 3, (1,), generator=g))))
        res = col(items)
        out["collate"].append(dict(n_frames_per_step=n_frames,
                                   items=[[x.tolist() if torch.is_tensor(x) else x for x in it] for it in items],
                                   result=[r.tolist() for r in res]))
    path = os.path.join(HERE, "data_golden.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes", len(out["sampler"]), "sampler cases")


if __name__ == "__main__":
    main()
