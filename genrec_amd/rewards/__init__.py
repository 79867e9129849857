"""Reward functions of the GRPO rollout (mirror of the reference's ``rewards`` package)."""

from .ranking_reward import (  # noqa: F401
    SidRewardScorer,
    build_reward_funcs,
    build_reward_setup,
    get_ndcg_rule_reward,
    get_prefix_rule_reward,
    rule_reward,
)
