cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in p1 p2; do
  echo "== $t events:"; VARPRO_HIP_LIBRARY=$R/varpro_amd/lib/ab/libvarpro_hip_$t.so timeout 200 python $R/tools/basis_var_probe.py 2>&1 | grep "pad        0" | cut -c1-75
  rm -rf /tmp/rp_$t; VARPRO_HIP_LIBRARY=$R/varpro_amd/lib/ab/libvarpro_hip_$t.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$t -o x -- python $R/tools/basis_var_probe.py > /dev/null 2>&1
  echo "   rocprof:"; grep -h "basis_rowpair" /tmp/rp_$t/*kernel_stats.csv /tmp/rp_$t/*/*kernel_stats.csv 2>/dev/null | head -2 | cut -c1-160
done
