"""CPU restatement of the dynamic-hint cascade's control flow.  TEST INFRASTRUCTURE:
only tests/, the smoke check and the CPU-baseline leg of the benchmark may import it.

Follows ``DynamicHintRuleOnlyGRPOTrainer._run_dynamic_hint_cascade``
(fixed_hint_grpo_trainer.py:558-693) with its helpers
``_resolve_example_dynamic_hint_max_depth`` (470-478), ``_build_runtime_hinted_example``
(480-490), ``_compute_rule_hit_rewards`` (33-44) and ``_has_global_unresolved_groups``
(511-521), on integer codes instead of text: an item is the list of numbers
``re.findall(r"_(\\d+)")`` extracts from its SID text, a hint of depth d is the
first d numbers of the ground truth (fixed_hint_utils.py:11-12), and a row hits
when hint + completion numbers equal the ground truth's.

Pinned against the reference itself: tests/golden/make_golden_cascade.py drives the
real method (with stand-ins for the trainer packages that are not installed) and
stores its decisions in tests/golden/cascade.json.gz.
"""

from __future__ import annotations

from typing import Callable, Optional, Sequence


def group_max_depth(overrides: Sequence[Optional[int]], default_max: int) -> int:
    """Largest per-example depth cap of a group (fixed_hint_grpo_trainer.py:470-478, 584-587)."""
    caps = []
    for override in overrides:
        caps.append(int(default_max) if override is None else max(min(int(override), int(default_max)), 0))
    return max(caps)


def rule_hits(gt: Sequence[int], hint_depth: int, completions: Sequence[Sequence[int]]) -> list[float]:
    """1.0 where hint numbers + completion numbers equal the ground truth (33-44)."""
    hint = list(gt[: max(hint_depth, 0)])
    return [1.0 if hint + list(c) == list(gt) else 0.0 for c in completions]


def run_cascade(
    gts: Sequence[Sequence[int]],
    overrides: Sequence[Sequence[Optional[int]]],
    num_generations: int,
    max_hint_depth: int,
    generate: Callable[[list[int], list[int]], list[list[list[int]]]],
    other_ranks_unresolved: Optional[Sequence[bool]] = None,
):
    """``gts[g]``: ground-truth numbers of group g; ``overrides[g][k]``: per-example depth caps.
    ``generate(group_ids, depths)`` returns, per listed group, ``num_generations`` completions
    (lists of numbers).  ``other_ranks_unresolved[s]`` says whether any OTHER rank still has
    unresolved groups after stage s (None: single process).

    Returns a dict with the selected completions / depths / hits per group, the per-stage
    statistics and the log of generate calls (groups and depths, dummy calls included).
    """
    n_groups = len(gts)
    caps = [group_max_depth(overrides[g], max_hint_depth) for g in range(n_groups)]
    unresolved = list(range(n_groups))
    selected: dict[int, dict] = {}
    stage_stats = []
    calls = []
    for depth in range(max_hint_depth + 1):
        local = list(unresolved)
        targets = local if local else [0]  # a rank with nothing left still generates for a dummy group (594-596)
        depths = [min(depth, caps[g]) for g in targets]
        outs = generate(list(targets), list(depths))
        calls.append({"groups": list(targets), "depths": list(depths), "dummy": not local})
        nxt = []
        hit_groups = 0
        picked = 0
        if local:
            for g, d, comps in zip(targets, depths, outs):
                rewards = rule_hits(gts[g], d, comps)
                hit_any = any(r > 0 for r in rewards)
                hit_groups += 1 if hit_any else 0
                if hit_any or depth == caps[g]:
                    # the stage index is stored as the group's hint depth (654), the prompt used d
                    selected[g] = {"completions": comps, "rule_rewards": rewards, "hint_depth": depth,
                                   "prompt_hint_depth": d, "rule_hit_any": hit_any}
                    picked += 1
                else:
                    nxt.append(g)
        stage_stats.append({"requested_hint_depth": depth, "evaluated_group_count": len(local),
                            "rule_hit_group_count": hit_groups, "selected_group_count": picked,
                            "remaining_group_count": len(nxt)})
        unresolved = nxt
        others = bool(other_ranks_unresolved[depth]) if other_ranks_unresolved is not None else False
        if not (len(unresolved) > 0 or others):
            break
    if len(selected) != n_groups:
        raise RuntimeError(f"Dynamic hint cascade selected {len(selected)} groups for {n_groups} prompt groups.")
    return {
        "selected": [selected[g] for g in range(n_groups)],
        "selected_group_hint_depths": [selected[g]["hint_depth"] for g in range(n_groups)],
        "selected_group_rule_hits": [selected[g]["rule_hit_any"] for g in range(n_groups)],
        "stage_stats": stage_stats,
        "calls": calls,
    }
