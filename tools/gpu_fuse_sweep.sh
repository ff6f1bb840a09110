cd $GRAFT_REPO_ROOT
A="--dtype f16x3 --timesteps 40 --steps 2 --warmup 1 --no-e2e-files --no-drift --no-configs4 --no-cpu-baseline --no-roofline --no-parity-mode"
for v in 100000 256 128 64 0; do
  echo -n "PRG_FUSE_PRO_MAX=$v streams 1: "; PRG_FUSE_PRO_MAX=$v python bench.py $A --streams 1 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step']/40)"
done
echo -n "streams 2: "; python bench.py $A --streams 2 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step']/40)"
echo -n "fp32 streams 1: "; python bench.py $A --dtype fp32 --streams 1 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step']/40)"
echo -n "fp32 streams 2: "; python bench.py $A --dtype fp32 --streams 2 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step']/40)"
