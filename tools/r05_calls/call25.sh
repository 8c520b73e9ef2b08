cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final_r05
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest_gpu_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
B="python $R/bench.py"
timeout 600 $B --steps 20 --warmup 5 > $OUT/bench_line_driver_command.json 2> $OUT/bench_line_driver_command.err
cd /tmp
CMD="$B --emulate-tp 8 --steps 8 --warmup 2 --tune 0 --profile-steps 0 --no-cpu-baseline --no-traffic --no-full-run"
TM_EMULATE_AR_GBPS=150 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t_overlap -o trace -- $CMD > $OUT/trace_overlap.log 2>&1
python $R/tools/rocpd_summary.py --overlap exchange_standin $OUT/t_overlap/trace_results.db > $OUT/prefill_overlap_kernel_trace.txt 2>&1
rm -rf $OUT/t_overlap
TM_EMULATE_AR_GBPS=150 TM_COMM_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t_serial -o trace -- $CMD > $OUT/trace_serial.log 2>&1
python $R/tools/rocpd_summary.py --overlap exchange_standin $OUT/t_serial/trace_results.db > $OUT/prefill_serial_kernel_trace.txt 2>&1
rm -rf $OUT/t_serial
cd $R
timeout 600 $B --model llama3_70b --quant-policy 4 --emulate-tp 8 --steps 128 --no-cpu-baseline --no-traffic --no-full-run > $OUT/bench_line_config3_llama3_70b_tp8_rank_emulation.json 2>/dev/null
timeout 600 $B --emulate-tp 8 --steps 128 --no-cpu-baseline --no-traffic --no-full-run > $OUT/bench_line_llama3_8b_tp8_rank_emulation.json 2>/dev/null
timeout 600 $B --emulate-tp 2 --steps 128 --no-cpu-baseline --no-traffic --no-full-run > $OUT/bench_line_llama3_8b_tp2_rank_emulation.json 2>/dev/null
cat $OUT/prefill_overlap_kernel_trace.txt $OUT/prefill_serial_kernel_trace.txt | cut -c1-200
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_line_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d.get('ms_per_step'), d.get('step_roofline',{}).get('frac'), d.get('value_1k_out'), d.get('ttft_p50_ms'), d.get('roofline',{}).get('frac'))
    except Exception as e: print(f, 'no json', e)
PY
