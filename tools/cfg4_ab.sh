#!/bin/bash
# A/B of Gram-fit builds on one box: tools/cfg4_ab.sh lib1.so lib2.so ...   (cfg4 at B = 8192 and B = 512 each)
for rnd in 1 2 3; do
for lib in "$@"; do
  echo "== $lib"
  VARPRO_HIP_LIBRARY=$lib python tools/cfg4_hist.py 8192 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=8192 ms %.3f mean %.2f max %d' % (d['ms'], d['mean'], d['max']))"
done; done
