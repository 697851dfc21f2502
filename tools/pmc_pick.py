"""Print (space-separated) the counters of a wish list that `rocprofv3 -L` lists on this GPU, at most `limit` of them.
usage: rocprofv3 -L > list.txt; python tools/pmc_pick.py list.txt LIMIT NAME [NAME ...]"""
import re
import sys

text = open(sys.argv[1], errors="replace").read()
have = set(re.findall(r"\b([A-Z][A-Za-z0-9_]{3,})\b", text))
out = [n for n in sys.argv[3:] if n in have][:int(sys.argv[2])]
print(" ".join(out))
