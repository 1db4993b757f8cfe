cd "${GRAFT_REPO_ROOT:-/root/repo}"
# 20-step regions over an epoch boundary (the driver's flags) with and without the event engine.step() records behind every replay
for rep in 1 2 3 4; do
  echo "== SRH_STEP_FLUSH=0"; SRH_STEP_FLUSH=0 STEP_EVENTS=0 REPEAT=2 bash tools/driver20_repeat.sh | cut -c1-140
  echo "== SRH_STEP_FLUSH=1 (default)"; SRH_STEP_FLUSH=1 STEP_EVENTS=0 REPEAT=2 bash tools/driver20_repeat.sh | cut -c1-140
done
