export TMPDIR=/tmp
cp ds2i_amd/libds2i_hip.so /tmp/orig.so
for O in 6 7 8; do
  cp ds2i_amd/csrc/build/variants/lib_occ$O.so ds2i_amd/libds2i_hip.so
  echo "== OCC $O"
  python bench.py --workload gov2 --steps 10 --warmup 2 --no-oracle 2>&1 | grep -E "^class [01]|^\{" | cut -c1-130
done
cp /tmp/orig.so ds2i_amd/libds2i_hip.so
