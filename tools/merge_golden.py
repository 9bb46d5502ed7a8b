"""python tools/merge_golden.py KEY FILE: put the JSON object in FILE (the last line of a generator's output) under
tests/golden/hotpath_golden.json[KEY] (generators that run on the GPU box print their result instead of writing it)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
key, src = sys.argv[1], sys.argv[2]
obj = json.loads(open(src).read().strip().splitlines()[-1])
path = os.path.join(ROOT, "tests", "golden", "hotpath_golden.json")
gold = json.load(open(path))
gold[key] = obj
with open(path, "w") as f:
    json.dump(gold, f, indent=1)
print("merged", key, "from", src)
