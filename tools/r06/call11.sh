#!/bin/bash
# round 6, call 11: the driver's default bench command with the new line (roofline per configuration, cpu_reference for C2)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06m; mkdir -p $O
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/time.txt
tail -3 $O/bench.err; cat $O/time.txt; wc -c $O/bench.json
