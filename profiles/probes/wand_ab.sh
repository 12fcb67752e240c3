# A/B of the disjunctive kernels on the GOV2-scale batch (GPU box): bash profiles/probes/wand_ab.sh
export TMPDIR=/tmp
b() { echo "== $1"; shift; env "$@" python bench.py --workload gov2 --op wand --steps 6 --warmup 2 --no-oracle 2>&1 | grep -E "^class [0-3]|^\{" | cut -c1-330 | sed 's/"unit".*"ms_per_step"/ ms_per_step/'; }
b default A=1
b no_bmw DS2I_NO_BMW_PRUNE=1
b scale_x2 DS2I_DISJ_SCALE=8,4,8,8
b scale_half DS2I_DISJ_SCALE=2,1,2,2
