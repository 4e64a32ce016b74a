#!/usr/bin/env python3
"""Regenerates tools/skip_hook.patch (the ORBX_SKIP development hook of tools/saturated.sh) against the current sources:
applies the edits in place, writes `git diff` of csrc/ to the patch file and restores the sources.  Run from the repo root
with a clean orb_slam2_e_amd/csrc."""
import subprocess, sys

if subprocess.run(["git", "diff", "--quiet", "--", "orb_slam2_e_amd/csrc"]).returncode != 0:
    sys.exit("orb_slam2_e_amd/csrc has uncommitted changes: this script restores it from git when it is done -- commit or stash first")

def edit(path, pairs):
    s = open(path).read()
    for old, new in pairs:
        assert s.count(old) == 1, (path, old[:80], s.count(old))
        s = s.replace(old, new)
    open(path, "w").write(s)

hook = ('    static int skipmask = getenv("ORBX_SKIP") ? atoi(getenv("ORBX_SKIP")) : 0; static int ncalls = 0; ++ncalls;\n')
edit("orb_slam2_e_amd/csrc/orbx_extract.hip", [
    ("    const int nl = ex->nlevels;\n    orbx::KernelProfiler &pf = ex->prof;\n",
     "    const int nl = ex->nlevels;\n    orbx::KernelProfiler &pf = ex->prof;\n" + hook + "    const int SK = ncalls < 8 ? 0 : skipmask;\n"),
    ("            hipLaunchKernelGGL(k_pyr_level0_lin,", "            if (!(SK & 1)) hipLaunchKernelGGL(k_pyr_level0_lin,"),
    ("            hipLaunchKernelGGL(k_pyr_level0, g,", "            if (!(SK & 1)) hipLaunchKernelGGL(k_pyr_level0, g,"),
    ("        hipLaunchKernelGGL(k_pyr_resize, g,", "        if (!(SK & 2)) hipLaunchKernelGGL(k_pyr_resize, g,"),
    ("    pf.start(2, st);\n    if (ex->TS == 52)", "    pf.start(2, st);\n    if (SK & 4) {} else if (ex->TS == 52)"),
    ("    hipLaunchKernelGGL(k_octree, dim3(nl, batch)", "    if (!(SK & 8)) hipLaunchKernelGGL(k_octree, dim3(nl, batch)"),
    ("        if (S > 256)\n            hipLaunchKernelGGL(k_blur<true>", "        if (SK & 16) {} else if (S > 256)\n            hipLaunchKernelGGL(k_blur<true>"),
    ("        hipLaunchKernelGGL(k_describe, dim3(nchunks, batch)", "        if (!(SK & 32)) hipLaunchKernelGGL(k_describe, dim3(nchunks, batch)"),
])
edit("orb_slam2_e_amd/csrc/orbm_match.hip", [
    ("                              int th, float nnratio, int *best, int *second, int *idx, int *match12, int *nmatch)\n{\n",
     "                              int th, float nnratio, int *best, int *second, int *idx, int *match12, int *nmatch)\n{\n" + hook +
     "    if (ncalls >= 8 && (skipmask & 64)) return;\n"),
])
open("tools/skip_hook.patch", "w").write(subprocess.check_output(["git", "diff", "--", "orb_slam2_e_amd/csrc"], text=True))
if "--keep" not in sys.argv:
    subprocess.check_call(["git", "checkout", "--", "orb_slam2_e_amd/csrc"])
print("tools/skip_hook.patch written")
