"""pipeline(path, backend_config=TurbomindEngineConfig(tp=2, devices=[0, 0])) as ONE call on the one-GPU box (not a test module;
tests/test_gpu_tp.py::test_pipeline_tp2_single_call runs it under a hard timeout, in a fresh interpreter so that the HIP runtime
sees GPU_MAX_HW_QUEUES).  The caller never launches ranks, broadcasts ids or passes `rank=`: Pipeline starts rank 1 itself
(lmdeploy_amd/turbomind/tp_group.py), both ranks share cuda:0, RCCL is skipped for the duplicate device and the native P2P
communicator carries the all-reduces.  Checked against the UNSHARDED oracle on oracle-assembled weights: greedy tokens of a
static batch and of a continuous-batching run.  argv: tmp dir.  Prints one JSON line."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tm_oracle as o                                 # noqa: E402
from tests.test_gpu_checkpoint import _fabricate                  # noqa: E402


def main():
    tmp = sys.argv[1]
    from lmdeploy_amd import GenerationConfig, TurbomindEngineConfig, pipeline
    cfg = o.ModelConfig(hidden=512, layers=2, q_heads=8, kv_heads=4, head_dim=128, inter=1024, vocab=640, kv_bits=8,
                        rope=o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192), rms_eps=1e-5)
    rng = np.random.default_rng(21)
    w = _fabricate(tmp, 'llama', cfg, rng)
    prompts = [rng.integers(3, cfg.vocab, n).astype(np.int32).tolist() for n in (19, 5, 40)]
    N = 6
    pipe = pipeline(tmp, backend_config=TurbomindEngineConfig(model_format='awq', quant_policy=8, max_batch_size=3, session_len=128, tp=2,
                                                              devices=[0, 0], communicator='cuda-ipc'))   # the reference's name for its
    # in-house communicator (device_comm.cc:14-30) -> the native P2P communicator; with both ranks on one device RCCL is not tried at all
    g = GenerationConfig(max_new_tokens=N, ignore_eos=True)
    static = [r.token_ids for r in pipe(prompts, g)]
    cont = [r.token_ids for r in sorted(pipe.generate_continuous(prompts + prompts[:2], g), key=lambda r: r.index)]
    backend = pipe.engine.comm_backend
    info = pipe.engine.comm_info()
    pipe.close()
    # the unsharded oracle, teacher-forced with the engine's tokens; compare where its top-2 margin exceeds the tolerance
    om = o.OracleModel(cfg, w, batch=len(prompts), max_ctx=128)
    ids, lg = om.forward([np.asarray(p, np.int32) for p in prompts])
    mism, checked = 0, 0
    for s in range(N):
        top2 = np.sort(lg.astype(np.float32), -1)[:, -2:]
        safe = (top2[:, 1] - top2[:, 0]) > 1.5e-2     # logits of this small model are O(0.3); tp = 2 vs the unsharded sums differ by ~1e-3
        mism += int(np.sum(np.asarray([static[b][s] for b in range(len(prompts))])[safe] != ids[safe]))
        checked += int(safe.sum())
        ids, lg = om.forward([[int(static[b][s])] for b in range(len(prompts))])
    print(json.dumps(dict(ok=True, backend=backend, info=info, mismatch=mism, checked=checked,
                          cont_equals_static=all(cont[i] == static[i % 3] for i in range(5)), static=static)))


if __name__ == '__main__':
    main()
