"""Edge cases of the scoring path recorded from the REAL reference: ``compute_metrics`` on records
whose prediction lists differ in length (evaluate.py:137-143, 168-170) and the reward functions on
text whose numbers do not fit 32 bits (util.py:133-134 compares Python integers).

Run in the build container only (needs the reference checkout; it never ships):

    python tests/golden/make_golden_edge.py [--reference /root/reference]

Asserts oracle == reference on every case, then writes tests/golden/scoring_edge.json.gz.
"""

from __future__ import annotations

import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import load_reference, save_json_gz  # noqa: E402

from oracle import scoring as oscore  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    ref = load_reference(args.reference)
    rng = np.random.default_rng(91)

    def sid(codes):
        return "".join(f"<{'abcd'[i]}_{int(c)}>" for i, c in enumerate(codes))

    def records(lengths, hit_every=3):
        out = []
        for i, n in enumerate(lengths):
            tgt = sid(rng.integers(0, 4, size=3))
            preds = [sid(rng.integers(0, 4, size=3)) for _ in range(n)]
            if n and i % hit_every == 0:
                preds[int(rng.integers(0, n))] = tgt
            out.append({"predict": preds, "output": tgt})
        return out

    metric_cases = {
        "shorter_lists_later": records([10, 10, 4, 0, 10, 7, 1, 10, 3, 10, 10, 2]),
        "first_sample_short": records([3, 10, 10, 5, 10, 20, 1, 0, 10]),
        "first_sample_empty": records([0, 10, 5, 10, 3, 0, 10]),
        "first_sample_longest": records([50, 10, 20, 50, 5, 50, 1, 50], hit_every=2),
        "string_predictions": [{"predict": "<a_1><b_2>", "output": "<a_1><b_2>"}, {"predict": "", "output": "<a_3>"},
                               {"predict": ["<a_3>", "<a_4>"], "output": ["<a_4>"]}],
        "all_empty": records([0, 0, 0]),
    }
    metrics = {}
    for name, data in metric_cases.items():
        got = ref.evaluate.compute_metrics(data)
        assert got == oscore.ref_compute_metrics(data), name
        metrics[name] = {"data": data, "metrics": got}
        print(f"  metrics {name}: {got}")

    # rewards on numbers beyond int32 / int64
    K = 4
    big = [2**31 - 1, 2**31, 2**32 + 5, 2**63, 10**25, 7]
    gts, comps = [], []
    for g in range(6):
        gt = [int(big[int(i)]) for i in rng.integers(0, len(big), size=3)]
        for k in range(K):
            pred = list(gt)
            if k % 3 == 1:
                pred[int(rng.integers(0, 3))] = int(big[int(rng.integers(0, len(big)))])
            elif k % 3 == 2:
                pred = pred[: int(rng.integers(0, 3))]
            gts.append("".join(f"<x_{v}>" for v in gt))
            comps.append("".join(f"<x_{v}>" for v in pred))
    kw = dict(reward_model=[{"ground_truth": g} for g in gts])
    rule = ref.rewards.rule_reward(None, comps, None, **kw)
    pre = ref.rewards.get_prefix_rule_reward(True)(None, comps, None, **kw)
    ndcg = ref.rewards.get_ndcg_rule_reward(K)(None, comps, None, **kw)
    assert rule == oscore.ref_rule_reward(comps, gts)
    assert pre == oscore.ref_prefix_reward(comps, gts)
    assert ndcg == oscore.ref_ndcg_reward(comps, gts, K)
    assert 0 < sum(rule) < len(rule)
    save_json_gz(os.path.join(HERE, "scoring_edge.json.gz"),
                 {"metrics": metrics, "K": K, "gts": gts, "comps": comps, "rule": rule, "prefix": pre, "ndcg": ndcg})


if __name__ == "__main__":
    main()
