#!/bin/bash
# round-2 evidence, part B: the other BASELINE configurations and the request-stream (continuous batching) line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final_r02
mkdir -p $OUT
B="python $R/bench.py"
timeout 400 $B --quant-policy 4 --steps 256 --no-cpu-baseline --no-full-run > $OUT/bench_line_llama3_8b_int4kv.json 2>/dev/null
timeout 400 $B --quant-policy 0 --steps 128 --no-cpu-baseline --no-full-run > $OUT/bench_line_config1_fp16kv.json 2>/dev/null
timeout 600 $B --model internlm2_20b --batch 128 --steps 128 --no-cpu-baseline --no-full-run > $OUT/bench_line_config2_internlm2_20b_b128.json 2>/dev/null
timeout 600 $B --model llama3_70b --quant-policy 4 --emulate-tp 8 --steps 128 --no-cpu-baseline --no-traffic --no-full-run > $OUT/bench_line_config3_llama3_70b_tp8_rank_emulation.json 2>/dev/null
timeout 900 $B --model mixtral_8x7b --steps 128 --no-cpu-baseline --no-traffic --no-full-run > $OUT/bench_line_config5_mixtral_fp8_tp1.json 2>/dev/null
timeout 400 python $R/tools/bench_continuous.py 2>/dev/null | grep '"metric"' > $OUT/continuous_batching_line.json
for f in $OUT/bench_line_llama3_8b_int4kv.json $OUT/bench_line_config1_fp16kv.json $OUT/bench_line_config2_internlm2_20b_b128.json $OUT/bench_line_config3_llama3_70b_tp8_rank_emulation.json $OUT/bench_line_config5_mixtral_fp8_tp1.json; do
  tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['step_roofline']['frac'])"
done
cat $OUT/continuous_batching_line.json | cut -c1-300
