#!/bin/bash
# GPU box, repo root: builds tools/cxx/one_frame_latency.cpp against the in-tree library and runs it
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
g++ -O2 -std=c++14 -D__HIP_PLATFORM_AMD__ -I $R/include -I /opt/rocm/include $R/tools/cxx/one_frame_latency.cpp -o /tmp/one_frame_latency -L $R/orb_slam2_e_amd -lorbslam_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$R/orb_slam2_e_amd -Wl,-rpath,/opt/rocm/lib && /tmp/one_frame_latency
