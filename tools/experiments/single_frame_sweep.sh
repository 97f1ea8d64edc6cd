#!/bin/bash
# Single-frame latency (tools/time_single_frame.py) over the small-batch switches: tile width, quadtree branches, graph replay.
export ORBX_STRICT_ENV=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tw in 72 108 144 288; do for br in 1 2 4; do
  echo -n "tile_w=$tw branches=$br: "; ORBX_TILE_W=$tw ORBX_SMALL_QT_BRANCHES=$br python tools/time_single_frame.py 2>/dev/null | grep "per call" | sed 's/batch 1: //' | tr '\n' ' '; echo
done; done
for br in 1 2 4; do echo -n "graph replay (branches $br): "; ORBX_GRAPH=1 ORBX_SMALL_QT_BRANCHES=$br python tools/time_single_frame.py 2>/dev/null | grep "per call" | tr '\n' ' '; echo; done
echo -n "batch schedule: "; ORBX_GRAPH=0 ORBX_QT_PER_LEVEL=0 python tools/time_single_frame.py 2>/dev/null | grep "per call" | tr '\n' ' '; echo
