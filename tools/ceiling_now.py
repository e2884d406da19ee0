"""Prints frieda_ctx_blake2s_ceiling_ex (rates, in-kernel clock, cycles per wave-compression) a few times; measurement aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frieda_amd
ctx = frieda_amd.Context(0)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    print({k: round(v / 1e9, 3) if k.endswith("per_s") else round(v, 3) for k, v in ctx.blake2s_ceiling_ex().items()}, ctx.blake2s_ceiling())
