"""Exploratory learning runs on the chase task (tests/learning_runs.py): python tools/explore_chase.py name[,name...] frames"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import learning_runs as LR
names = sys.argv[1].split(",")
frames = int(float(sys.argv[2]))
CFG = {"dqn": ("dqn", {}, "Breakout"), "rainbow": ("c51", LR.RAINBOW, "Breakout"), "iqn": ("iqn", {}, "Asterix"), "fqf": ("fqf", {}, "Asterix"),
       "dqn_n3": ("dqn", {"learner.n_step_q": 3}, "Breakout"), "dqn_flat": ("dqn", {"replay.policy": "prioritize", "replay.sumtree": "false"}, "Breakout")}
for nm in names:
    sab = None
    base = nm
    if ":" in nm:
        base, sab = nm.split(":")
    algo, extra, env_id = CFG[base]
    launch = base.endswith("_lp")
    r = LR.run(algo, extra, frames, launch, env_id=env_id, task="chase", sabotage=sab)
    print(nm, json.dumps({k: v for k, v in r.items() if k != "curve"}), flush=True)
    print("   curve:", r["curve"], flush=True)
