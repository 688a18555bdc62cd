#!/usr/bin/env python3
"""Known answers of CPython's MT19937 beyond the first state regeneration (draw 312): random.seed(s) followed by 700
random.random() calls, as hex floats.  Uses only the standard library (the reference calls exactly these:
init.py:137, 139; lib.py:434).  Writes tests/golden/kat_rng_long.json."""
import json
import os
import random

SEEDS = [1, 74703609, 2 ** 32 - 1, 2 ** 32 + 5, 2 ** 64 - 1]
out = {}
for s in SEEDS:
    random.seed(s)
    out[str(s)] = [random.random().hex() for _ in range(700)]
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_rng_long.json"), "w"))
print("wrote", len(out), "seeds")
