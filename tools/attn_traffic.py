#!/usr/bin/env python
"""HBM traffic of the decode attention kernel from a rocprofv3 --pmc FETCH_SIZE pass over
`bench.py --steps K --warmup W --profile-steps 0 --no-graph`: mean FETCH_SIZE per launch, corrected as
/opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950 (the counter reports KiB and exactly half of
a wide coalesced streaming read: x 1024 x 2), next to the algorithmic KV bytes of the same launches.
usage: attn_traffic.py <pmc.db> batch prompt_len warmup steps   -> one JSON object on stdout"""
import json
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    B, S, W, K = map(int, sys.argv[2:6])
    rows = list(db.execute("select avg(value), count(*) from counters_collection where kernel_name like "
                           "'%decode_attention%' and counter_name = 'FETCH_SIZE'"))
    mean_kib, n = rows[0]
    steps = W + K
    ctx_mean = S + 1 + (steps - 1) / 2.0                 # k_len of decode step j is S + 1 + j
    kv_bytes_per_token_layer = 2 * 8 * (128 + 4)          # Llama-3-8B int8 KV: K and V, 8 kv heads, 128 B data + 4 B params
    algorithmic = B * ctx_mean * kv_bytes_per_token_layer
    traffic = mean_kib * 1024.0 * 2.0
    print(json.dumps({'kernel': 'decode_attention_i8_mfma_kernel', 'launches': n, 'fetch_size_kib_mean': mean_kib,
                      'traffic_bytes_per_launch': traffic, 'algorithmic_bytes_per_launch': algorithmic,
                      'traffic_over_algorithmic': traffic / algorithmic, 'ctx_mean': ctx_mean,
                      'correction': 'FETCH_SIZE [KiB] x 1024 x 2 (gfx950 rocprofv3 reports half of a wide streaming read)'}))


if __name__ == '__main__':
    main()
