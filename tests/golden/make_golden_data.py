"""Generates tests/golden/data_golden.json from the REAL reference sampler
(/root/reference/datasets/sampler.py, imported as-is) and from Pillow's
Image.resize(NEAREST) (the call the reference's joint transforms make on label
maps).  Run in the build container: python tests/golden/make_golden_data.py"""
import importlib.util
import json
import os
import sys

sys.dont_write_bytecode = True       # /root/reference is read-only input: leave no __pycache__ behind in it

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_sampler", "/root/reference/datasets/sampler.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {"sampler": [], "nearest": []}
for n, world, pad, cons, perm, epoch in [(10, 3, True, False, True, 0), (10, 3, False, False, True, 5),
                                         (2975, 8, True, False, True, 17), (2975, 16, False, True, False, 3),
                                         (7, 8, True, False, False, 1), (500, 4, True, True, True, 99),
                                         (1, 2, True, False, True, 2)]:
    for rank in sorted({0, world // 2, world - 1}):
        s = ref.DistributedSampler(list(range(n)), pad=pad, consecutive_sample=cons, permutation=perm,
                                   num_replicas=world, rank=rank)
        s.set_epoch(epoch)
        out["sampler"].append({"n": n, "world": world, "rank": rank, "pad": pad, "consecutive": cons,
                               "permutation": perm, "epoch": epoch, "indices": [int(i) for i in s]})

rng = np.random.default_rng(7)
for (hs, ws), (hd, wd) in [((16, 24), (7, 9)), ((33, 47), (66, 95)), ((1024, 2048), (37, 51)),
                           ((210, 376), (656, 476)), ((170, 552), (317, 347)), ((19, 23), (19, 23)),
                           ((5, 3), (64, 80))]:
    m = rng.integers(0, 256, (hs, ws), dtype=np.uint8)
    r = np.array(Image.fromarray(m).resize((wd, hd), Image.NEAREST))
    # the source masks are reproducible from the seed (drawn in this order): store checksums, not pixels
    out["nearest"].append({"src": [hs, ws], "dst": [hd, wd], "seed_order": len(out["nearest"]),
                           "sum": int(r.astype(np.int64).sum()),
                           "weighted": int((r.astype(np.int64) * (np.arange(r.size).reshape(r.shape) % 251 + 1)).sum()),
                           "first_row": r[0].tolist()[:64], "last_col": r[:, -1].tolist()[:64]})
with open(os.path.join(HERE, "data_golden.json"), "w") as f:
    json.dump(out, f)
print("wrote", len(out["sampler"]), "sampler cases,", len(out["nearest"]), "resize cases")
