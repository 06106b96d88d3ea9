#!/bin/bash
# same entry point as the reference script/paraA/run.sh: all five operations on 4 clusters
python3 "$(dirname "$0")/../sweep.py" --set A --cluster "${1:-4}"
