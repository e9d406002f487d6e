#!/bin/bash
# round 4, run 11: full evidence collection at the current state (tools/collect_profiles.sh r04)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
bash tools/collect_profiles.sh r04
