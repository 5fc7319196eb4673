#!/bin/bash
# final library of round 5: the profiles (tools/profile_bench.sh); bench lines in a second call, after tools/summarize_profile.py
bash tools/profile_bench.sh
