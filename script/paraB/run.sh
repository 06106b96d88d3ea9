#!/bin/bash
# same entry point as the reference script/paraB/run.sh: all five operations on 4 clusters
python3 "$(dirname "$0")/../sweep.py" --set B --cluster "${1:-4}"
