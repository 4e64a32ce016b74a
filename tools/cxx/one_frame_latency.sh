#!/bin/bash
# GPU box, repo root: builds tools/cxx/one_frame_latency.cpp against the in-tree library and runs it
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
g++ -O2 -std=c++14 -I $R/include $R/tools/cxx/one_frame_latency.cpp -o /tmp/one_frame_latency -L $R/orb_slam2_e_amd -lorbslam_hip -Wl,-rpath,$R/orb_slam2_e_amd && /tmp/one_frame_latency
