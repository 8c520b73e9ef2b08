cd $GRAFT_REPO_ROOT
for pad in 0 24576 0 24576; do
  echo "== TM_ATTN_LDS_PAD=$pad"
  TM_ATTN_LDS_PAD=$pad timeout 200 python tools/bench_attention.py --ctx 1040 --splits 1 --layers 32 --iters 40 2>&1 | grep ctx=
  TM_ATTN_LDS_PAD=$pad timeout 200 python tools/bench_attention.py --ctx 1536 --splits 1 --layers 32 --iters 40 2>&1 | grep ctx=
done
