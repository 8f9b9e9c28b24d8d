#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/suite_repeat.sh 2 r03_suite_repeat_last_two
