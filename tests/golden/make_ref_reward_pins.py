"""Extracts the only numeric pin the reference tree holds for the step path: unshaped env
rewards logged by v1_multiagent_ppo.py (sum per agent per episode) in
/root/reference/v1_final_results_MultiGrid-Cluttered-Fixed-15x15_3agents_1749607927.json.

Every per-agent episode total below 1.0 is written, sorted and de-duplicated, to
ref_v1_env_reward_values.json.  A total below 2*min(_reward) = 0.2 can only be a single goal
touch, i.e. exactly one value of MiniGridEnv._reward() = 1 - 0.9*(step_count/max_steps).
Run here (the reference is read as DATA; nothing is imported from it):
    python tests/golden/make_ref_reward_pins.py
"""
import json
import os

SRC = '/root/reference/v1_final_results_MultiGrid-Cluttered-Fixed-15x15_3agents_1749607927.json'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ref_v1_env_reward_values.json')

with open(SRC) as f:
  d = json.load(f)
vals = set()
for ep in d['episode_rewards']:
  for v in ep:
    if 0 < v < 1.0:
      vals.add(float(v))
vals = sorted(vals)
with open(OUT, 'w') as f:
  json.dump({'source': os.path.basename(SRC), 'key': 'episode_rewards', 'max_steps': 100,
             'values_below_1': [repr(v) for v in vals]}, f, indent=0)
print(len(vals), 'values ->', OUT)
