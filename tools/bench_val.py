"""Reads bench.py's JSON line from stdin and prints "<Mrays/s> <ms per step>" (for sweep scripts)."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"], 2))
