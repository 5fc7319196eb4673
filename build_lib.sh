#!/bin/bash
# Builds rtiow_amd/librtiow_hip.so (gfx950) and oracle/liboracle.so: same as __graft_entry__.build().
set -e
cd "$(dirname "$0")"
python3 -c "import __graft_entry__ as g; g.build(); print('built')"
