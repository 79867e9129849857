"""Time of the token-level prefix scoring (grx_token_prefix_advantages) for one rank's rollout shard
(8192 rows, 16 per group, 6-token completions), beside the CPU restatement of the reference's per-row loops."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genrec_amd.token_advantage import pack_ground_truth_ids, token_prefix_advantages  # noqa: E402
from oracle import scoring as oscore  # noqa: E402  (the CPU leg of this measurement only)


def main():
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    K, groups, levels, Tc, eos = 16, 512, 4, 6, 2
    R = K * groups
    gts, ids, mask = [], np.zeros((R, Tc), dtype=np.int64), np.zeros((R, Tc), dtype=np.int64)
    for g in range(groups):
        gt = [1000 + 256 * l + int(rng.integers(0, 4)) for l in range(levels)]
        for k in range(K):
            r = g * K + k
            gts.append(gt)
            depth = int(rng.integers(0, levels + 1))
            toks = list(gt[:depth]) + [1000 + 256 * l + int(rng.integers(0, 4)) for l in range(depth, levels)] + [eos]
            ids[r, : len(toks)] = toks
            mask[r, : len(toks)] = 1
    tids, tmask = torch.from_numpy(ids), torch.from_numpy(mask)
    ndcg = torch.from_numpy((rng.random(R) * -0.4).astype(np.float32))
    gt_ids, gt_len = pack_ground_truth_ids(gts, dev)
    d_ids, d_mask, d_ndcg = tids.to(dev), tmask.to(dev), ndcg.to(dev)
    for total, pen in ((False, False), (True, False), (True, True)):
        kw = dict(has_ndcg=True, eos_token_id=eos, token_adv_total_token_normalize=total,
                  token_level_ndcg_error_token_penalty=pen)
        for _ in range(5):
            token_prefix_advantages(d_ids, d_mask, gt_ids, gt_len, K, 1.0, 0.5, d_ndcg, **kw)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        n = 200
        for _ in range(n):
            token_prefix_advantages(d_ids, d_mask, gt_ids, gt_len, K, 1.0, 0.5, d_ndcg, **kw)
        ev1.record()
        torch.cuda.synchronize()
        gpu_us = ev0.elapsed_time(ev1) / n * 1e3
        t0 = time.perf_counter()
        tr, sr, em = oscore.ref_prefix_token_rewards(tids, tmask, gts, eos, True, True)
        oscore.ref_token_prefix_advantages(tr, sr, em, tmask, K, 1.0, 0.5, True, ndcg, total, pen)
        cpu_s = time.perf_counter() - t0
        print(f"total_token_normalize={total} error_token_penalty={pen}: GPU {gpu_us:.1f} us per call "
              f"(host wrapper included), CPU restatement {cpu_s:.2f} s for the same {R} rows")


if __name__ == "__main__":
    main()
