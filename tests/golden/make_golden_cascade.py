"""Golden decisions of the dynamic-hint cascade, recorded from the REAL reference.

Run in the build container only (needs the reference checkout; it never ships):

    python tests/golden/make_golden_cascade.py [--reference /root/reference]

Drives ``DynamicHintRuleOnlyGRPOTrainer._run_dynamic_hint_cascade``
(fixed_hint_grpo_trainer.py:558-693) on randomized scenarios.  The trainer
packages that are not installed here (`trl`) get empty stand-ins, exactly as the
reference's own tests/test_dynamic_hint_trainer.py does; generation is replaced
by a table lookup (scenario data), everything that decides -- hint construction,
rule-hit rewards, group selection, stage statistics, the cross-rank "anyone
unresolved" gather -- is the reference's code.  The CPU restatement in
oracle/cascade.py is asserted equal on every scenario before the fixture
(inputs + the reference's decisions, integers only) is written.
"""

from __future__ import annotations

import argparse
import gzip
import importlib.util
import json
import os
import re
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import cascade as ocascade  # noqa: E402

LETTERS = "abcdefgh"


def load_trainer_module(ref_root: str):
    def stub(name, **attrs):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(mod, k, v)
            sys.modules[name] = mod

    stub("trl", GRPOTrainer=object, GRPOConfig=object)
    stub("trl.data_utils", is_conversational=lambda example: False,
         maybe_apply_chat_template=lambda data, processing_class: data)
    stub("trl.trainer", )
    stub("trl.trainer.grpo_trainer", nanstd=lambda tensor: tensor)
    stub("trl.trainer.utils", pad=lambda values, padding_value=None, padding_side=None: values)
    sys.path.insert(0, ref_root)
    spec = importlib.util.spec_from_file_location("ref_fixed_hint_grpo_trainer",
                                                  os.path.join(ref_root, "fixed_hint_grpo_trainer.py"))
    module = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = module
    spec.loader.exec_module(module)
    return module


def text_of(codes, start_level=0):
    return "".join(f"<{LETTERS[start_level + i]}_{int(c)}>" for i, c in enumerate(codes))


def numbers_of(text):
    return [int(m) for m in re.findall(r"_(\d+)", text)]


def make_scenario(rng, n_groups, K, L, max_depth, p_hit, with_overrides, n_ranks):
    gts = rng.integers(0, 12, size=(n_groups, L)).tolist()
    overrides = []
    for _ in range(n_groups):
        if with_overrides and rng.random() < 0.5:
            # per-example caps inside a group may differ; the group takes the largest
            overrides.append([None if rng.random() < 0.3 else int(rng.integers(-1, max_depth + 2)) for _ in range(K)])
        else:
            overrides.append([None] * K)
    table = {}  # (group, prompt depth) -> K completions of L - depth numbers
    for g in range(n_groups):
        for d in range(max_depth + 1):
            comps = []
            for _ in range(K):
                if rng.random() < p_hit * (1 + d):
                    comps.append(list(gts[g][d:]))
                else:
                    comps.append(rng.integers(0, 12, size=L - d).tolist())
            table[(g, d)] = comps
    others = None
    if n_ranks > 1:
        others = [bool(rng.random() < 0.6) for _ in range(max_depth + 1)]
    return dict(gts=gts, overrides=overrides, table=table, K=K, L=L, max_depth=max_depth, others=others)


def run_reference(module, sc):
    K, max_depth = sc["K"], sc["max_depth"]
    trainer = object.__new__(module.DynamicHintRuleOnlyGRPOTrainer)
    trainer.num_generations = K
    stage_no = [0]

    def gather(t):
        if sc["others"] is None:
            return t
        flag = 1 if sc["others"][stage_no[0]] else 0
        stage_no[0] += 1
        return torch.cat([t, torch.tensor([flag], dtype=t.dtype)])

    trainer.accelerator = types.SimpleNamespace(is_main_process=False, device="cpu", gather=gather)
    trainer._build_hinted_prompts = lambda inputs: [
        f"{ex['prompt']}|{ex['oracle_hint_depth']}|{ex['oracle_hint_text']}" for ex in inputs
    ]
    trainer.processing_class = types.SimpleNamespace(batch_decode=lambda batch, skip_special_tokens=True: batch)
    calls = []

    def generate(prompts_text, images):
        groups, depths, completions = [], [], []
        for i in range(0, len(prompts_text), K):
            g, d, hint = prompts_text[i].split("|")
            g, d = int(g[1:]), int(d)
            assert numbers_of(hint) == sc["gts"][g][:d]
            groups.append(g)
            depths.append(d)
            completions.extend(text_of(c, d) for c in sc["table"][(g, d)])
        calls.append({"groups": groups, "depths": depths})
        return prompts_text, completions

    trainer._generate_dynamic_stage = generate
    inputs = []
    for g, gt in enumerate(sc["gts"]):
        for k in range(K):
            ex = {"prompt": f"g{g}", "reward_model": {"ground_truth": text_of(gt)}}
            if sc["overrides"][g][k] is not None:
                ex["dynamic_hint_max_depth_override"] = sc["overrides"][g][k]
            inputs.append(ex)
    out = trainer._run_dynamic_hint_cascade(inputs, None, max_depth)
    n_groups = len(sc["gts"])
    return {
        "selected_group_hint_depths": [int(x) for x in out["selected_group_hint_depths"]],
        "selected_group_rule_hits": [bool(x) for x in out["selected_group_rule_hits"]],
        "selected_rule_rewards": [float(x) for x in out["selected_rule_rewards"]],
        "selected_completions": [numbers_of(t) for t in out["selected_completions_text"]],
        "selected_prompt_hint_depths": [int(out["selected_inputs"][g * K]["oracle_hint_depth"]) for g in range(n_groups)],
        "stage_stats": out["stage_stats"],
        "calls": calls,
    }


def run_oracle(sc):
    def generate(groups, depths):
        return [sc["table"][(g, d)] for g, d in zip(groups, depths)]

    return ocascade.run_cascade(sc["gts"], sc["overrides"], sc["K"], sc["max_depth"], generate, sc["others"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    module = load_trainer_module(args.reference)
    rng = np.random.default_rng(20260311)
    cases = []
    shapes = [(1, 2, 4, 0, 0.2, False, 1), (3, 2, 4, 3, 0.05, False, 1), (6, 4, 4, 3, 0.08, True, 1),
              (5, 3, 3, 2, 0.1, True, 2), (8, 4, 4, 3, 0.02, True, 2), (4, 2, 4, 3, 0.5, False, 2),
              (7, 16, 4, 3, 0.01, True, 1), (2, 2, 4, 1, 0.0, True, 2)]
    for shape in shapes:
        for _ in range(4):
            sc = make_scenario(rng, *shape)
            ref = run_reference(module, sc)
            orc = run_oracle(sc)
            K = sc["K"]
            assert orc["selected_group_hint_depths"] == ref["selected_group_hint_depths"]
            assert orc["selected_group_rule_hits"] == ref["selected_group_rule_hits"]
            assert [r for s in orc["selected"] for r in s["rule_rewards"]] == ref["selected_rule_rewards"]
            assert [c for s in orc["selected"] for c in s["completions"]] == ref["selected_completions"]
            assert [s["prompt_hint_depth"] for s in orc["selected"]] == ref["selected_prompt_hint_depths"]
            assert orc["stage_stats"] == ref["stage_stats"]
            assert [{"groups": c["groups"], "depths": c["depths"]} for c in orc["calls"]] == ref["calls"]
            cases.append({
                "K": K, "L": sc["L"], "max_depth": sc["max_depth"], "gts": sc["gts"], "overrides": sc["overrides"],
                "others": sc["others"],
                "table": [[g, d, comps] for (g, d), comps in sorted(sc["table"].items())],
                "expected": ref,
            })
    path = os.path.join(HERE, "cascade.json.gz")
    with gzip.open(path, "wt", encoding="utf-8") as f:
        json.dump({"cases": cases}, f, separators=(",", ":"))
    print(f"wrote {os.path.relpath(path, ROOT)} ({os.path.getsize(path)} bytes), {len(cases)} scenarios; "
          f"oracle == reference on all of them")


if __name__ == "__main__":
    main()
