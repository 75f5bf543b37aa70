#!/bin/bash
out=gpurun_out/r05k; mkdir -p $out; rm -f $out/*.log
sets=$(python3 -c "
import itertools
print(';'.join('pr.n_class_order=' + str(int(''.join(map(str,p)))) for p in itertools.permutations(range(4))))")
for cfg in "1048576 5000000 1" "1048576 5000000 2" "10000000 50000000 1" "10000000 50000000 2"; do
  set -- $cfg
  N=$1 E=$2 K=$3 R=3 OPTSETS="$sets" timeout -k 10 300 python tools/pr_exp.py 2>&1 | grep lib= | sed 's/lib=product //; s/probe \[\]//' | sort -t' ' -k8 -n | head -30 >> $out/norder.log
done
cat $out/norder.log | awk '{print $1,$2,$3,$4,$7,$8}' | column -t
