# BASELINE configs[4] on ONE GPU, game by game: the reference's 8-game suite configuration (README.md:62-112 atari8_double_duel_prior: fqf + double-Q +
# dueling + prioritized replay) at full size, one bench line per game (action sets of 4 / 6 / 9 / 18).  The 8-GPU data-parallel form of the same
# configuration is what the driver's scaling run measures; this is the per-rank workload of each game.  Output: gpurun_out/<round>/suite8.json
cd $GRAFT_REPO_ROOT; R=${R:-r03}
mkdir -p gpurun_out/$R
echo "[" > gpurun_out/$R/suite8.json
sep=""
for game in Asterix BeamRider Breakout Enduro MsPacman Qbert Seaquest SpaceInvaders; do
  line=$(python3 bench.py --no-cpu-baseline --no-ratio320 --no-other-entry --steps 3 --warmup 2 --algo fqf --env $game learner.double_q=true learner.dueling_head=true replay.policy=prioritize 2> gpurun_out/$R/suite8_$game.err | tail -1)
  echo "$sep$line" >> gpurun_out/$R/suite8.json; sep=","
  echo "$line" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$game', d['config'].get('actions', d['config']), d['value'], d['ms_per_step'], d['updates_per_sec'], d['last_loss'])" | cut -c1-200
done
echo "]" >> gpurun_out/$R/suite8.json
