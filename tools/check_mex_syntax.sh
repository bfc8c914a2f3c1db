#!/bin/bash
# Type-checks the MEX gateways against include/ofdm_mi355x.h without MATLAB (syntax only, nothing is linked).
cd "$(dirname "$0")/.."
rc=0
for f in ofdm-course_amd/mex/*.cpp; do
  g++ -std=c++17 -fsyntax-only -Wall -Wextra -Wno-unused-parameter -Itools/mex_syntax -Iinclude "$f" || rc=1
done
exit $rc
