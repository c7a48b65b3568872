#!/bin/bash
# bench lines + kernel trace of the default command only (the library is unchanged: the PMC passes stay valid)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/refresh; mkdir -p $O/lines; cd $R
timeout -k 10 400 python bench.py > $O/lines/n1.json 2> $O/lines/err.txt || exit 1
timeout -k 10 300 python bench.py --no-tables-ahead --no-cpu-baseline > $O/lines/n1_one_stream_two_handles.json 2>> $O/lines/err.txt || exit 1
bash tools/profile_bench.sh
tail -1 $O/lines/n1.json | cut -c1-300
