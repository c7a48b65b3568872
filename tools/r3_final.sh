#!/bin/bash
# everything profiles/ quotes, on the library as it is now: the refresh, then the default-loop lines / sampling table on top
bash tools/refresh_profiles.sh > gpurun_out/refresh_log.txt 2>&1 || { tail -5 gpurun_out/refresh_log.txt; exit 1; }
bash tools/r3_lines.sh > /dev/null 2>&1 || exit 1
bash tools/r3_table1.sh > /dev/null 2>&1
tail -3 gpurun_out/refresh/progress.txt
