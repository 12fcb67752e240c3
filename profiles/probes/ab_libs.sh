# A/B over library builds kept under profiles/tmp_libs (DS2I_EXTRA_CFLAGS variants): bash profiles/probes/ab_libs.sh name...
export TMPDIR=/tmp
cp ds2i_amd/libds2i_hip.so /tmp/orig.so
for v in default "$@"; do
  if [ "$v" != default ]; then cp profiles/tmp_libs/lib_$v.so ds2i_amd/libds2i_hip.so; else cp /tmp/orig.so ds2i_amd/libds2i_hip.so; fi
  python bench.py --workload gov2 --steps 12 --warmup 2 --no-oracle 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'qps', round(d['value']), 'ms/step', round(d['ms_per_step'],2), d['step_ms_spread'], 'resident', round(d.get('kernel_resident_qps',0)), [round(c['ms_per_launch'],2) for c in d['roofline']['per_class']])
"
done
cp /tmp/orig.so ds2i_amd/libds2i_hip.so
