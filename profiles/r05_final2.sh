#!/bin/bash
# final tree: counter passes + kernel tables (r05_pmc.sh), then tests, smoke, bench lines (r05_final.sh)
cd "$(dirname "$0")/.."
bash profiles/r05_pmc.sh 2>&1 | grep -E "^==|tile_adam|bin_kernel|mean of last|threads" | cut -c1-150
bash profiles/r05_final.sh
