"""The DATA the hot path shares with the reference -- the rBRIEF pattern, patch / border sizes, matcher thresholds, the FEM call's material
and penalty constants, the Levenberg hook's weights and the outlier thresholds -- read as text from the reference's sources where they lie
and compared with what the oracle and the product carry.  Dev container only (the reference mount does not travel): skipped elsewhere.
This pins tables and constants, not algorithms: the oracle's arithmetic stays "parity unpinned" (DESIGN 5)."""
import inspect
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference mount not present")


def _ref(path):
    return open(os.path.join(REF, path), errors="replace").read()


def _ours(path):
    return open(os.path.join(ROOT, path)).read()


def _nocomment(t):
    return re.sub(r"//[^\n]*", " ", re.sub(r"/\*.*?\*/", " ", t, flags=re.S))


def _ints(text):
    return [int(v) for v in re.findall(r"-?\d+", _nocomment(text))]


def test_rbrief_pattern_is_the_references_table():
    src = _ref("src/ORBextractor.cc")
    body = src[src.index("bit_pattern_31_[256*4]"):]
    ref = _ints(body[body.index("{") + 1: body.index("};")])
    assert len(ref) == 1024
    prod = _ints(_ours("orb_slam2_e_amd/csrc/orb_pattern.inc"))
    assert prod == ref
    o = _ours("oracle/orb_pattern_data.h")
    assert _ints(o[o.index("{") + 1: o.index("};")]) == ref


def _const(text, name):
    m = re.search(r"\b" + name + r"\s*=\s*([-0-9.eE]+)", _nocomment(text))
    assert m, name
    return float(m.group(1))


def _define(text, name):
    m = re.search(r"#define\s+" + name + r"\s+([-0-9.eE]+)", text)
    assert m, name
    return float(m.group(1))


def test_extractor_sizes():
    ref, orc = _ref("src/ORBextractor.cc"), _ours("oracle/orb_oracle.c")
    for name in ("PATCH_SIZE", "HALF_PATCH_SIZE", "EDGE_THRESHOLD"):
        assert _const(ref, name) == _define(orc, name), name
    m = re.search(r"constexpr int EDGE\s*=\s*(\d+)", _ours("orb_slam2_e_amd/csrc/orbx_internal.h"))
    assert m and float(m.group(1)) == _const(ref, "EDGE_THRESHOLD")


def test_matcher_thresholds_are_this_forks():
    """This fork lowers upstream's 100 / 50 to 95 / 45 (ORBmatcher.cc:37-38)."""
    ref = _ref("src/ORBmatcher.cc")
    th_high, th_low, histo = (_const(ref, "ORBmatcher::" + n) for n in ("TH_HIGH", "TH_LOW", "HISTO_LENGTH"))
    assert (th_high, th_low, histo) == (95, 45, 30)
    orc = _ours("oracle/match_oracle.c")
    assert _define(orc, "HISTO_LENGTH") == histo
    assert {float(v) for v in re.findall(r"\bTH_LOW\s*=\s*(\d+)", orc)} == {th_low}
    st = _ours("oracle/stereo_oracle.c")
    assert float(re.search(r"TH_HIGH\s*=\s*(\d+)", st).group(1)) == th_high and float(re.search(r"TH_LOW\s*=\s*(\d+)", st).group(1)) == th_low
    # the host mirror and the shells pass the same two numbers through the C-ABI's th / th_high arguments
    from orb_slam2_e_amd.matcher import ORBmatcher
    assert (ORBmatcher.TH_HIGH, ORBmatcher.TH_LOW, ORBmatcher.HISTO_LENGTH) == (th_high, th_low, histo)


def test_fem_call_constants():
    """Optimizer.cc:480 builds FEA2(id, E, nu, h, fg, nElType, debug); FEA2.cc:105-107 imposes the encastre with Klarge."""
    m = re.search(r"FEA2\s+fea2\(\s*pFrame->mnId\s*,\s*([-0-9.eE]+)\s*,\s*([-0-9.eE]+)\s*,\s*([-0-9.eE]+)\s*,\s*([-0-9.eE]+)\s*,", _nocomment(_ref("src/Optimizer.cc")))
    assert m
    E, nu, h, fg = (float(v) for v in m.groups())
    klarge = {float(v) for v in re.findall(r"ImposeDirichletEncastre_K\(\s*nMode\s*,\s*\w+\s*,\s*([-0-9.eE]+)\s*\)", _nocomment(_ref("Thirdparty/g2o/g2o/FEA/src/FEA2.cc")))}
    assert len(klarge) == 1
    from orb_slam2_e_amd.fem import FEA2
    import oracle
    d = {k: v.default for k, v in inspect.signature(FEA2.__init__).parameters.items()}
    assert (d["E"], d["nu"], d["fg"]) == (E, nu, fg)
    assert inspect.signature(FEA2.ImposeDirichletEncastre_K).parameters["Klarge"].default == klarge.pop()
    od = {k: v.default for k, v in inspect.signature(oracle.fem_ke).parameters.items()}
    assert (od["E"], od["nu"], od["fg"]) == (E, nu, fg)
    smoke = _ours("tests/cxx/dropin_smoke.cpp")
    m2 = re.search(r"FEA2\s+fea\(\s*([-0-9.eE]+)\s*,\s*([-0-9.eE]+)f?\s*,\s*([-0-9.eE]+)f?\s*,\s*([-0-9.eE]+)f?\s*,", smoke)
    assert m2 and tuple(float(v) for v in m2.groups()) == (E, nu, h, fg)


def test_levenberg_hook_and_outlier_constants():
    """optimization_algorithm_levenberg.cpp: the hook's weights (w_rE 1, w_sE 5, first trial 1 / 2), ten trials after a failure, the
    good-step scales; Optimizer.cc: four rounds of ten iterations, chi2 5.991 for every round."""
    lev = _nocomment(_ref("Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp"))
    orc = _nocomment(_ours("oracle/pose_nr_oracle.c"))
    for pat in (r"w_rE\s*=\s*([0-9.]+)", r"w_sE\s*=\s*([0-9.]+)"):
        assert sorted(float(v) for v in re.findall(pat, lev)) == sorted(float(v) for v in re.findall(pat, orc)), pat
    assert float(re.search(r"_maxTrialsAfterFailure\s*=[^;]*?(\d+)\s*\)?\s*;", lev).group(1)) == float(re.search(r"_maxTrialsAfterFailure\s*=\s*(\d+)", orc).group(1)) == 10
    opt = _nocomment(_ref("src/Optimizer.cc"))
    fn = opt[opt.index("Optimizer::PoseOptimizationNR"):]
    fn = fn[:fn.index("\nint Optimizer::", 10)] if "\nint Optimizer::" in fn[10:] else fn
    its = re.search(r"its\[4\]\s*=\s*\{([^}]*)\}", fn)
    assert its and _ints(its.group(1)) == _ints(re.search(r"its\[4\]\s*=\s*\{([^}]*)\}", orc).group(1)) == [10, 10, 10, 10]
    chi = re.search(r"chi2\[4\]\s*=\s*\{([^}]*)\}", fn)
    assert chi and {float(v) for v in re.findall(r"[0-9.]+", chi.group(1))} == {5.991}
    g2o = _ours("oracle/mini_g2o.h")
    assert {float(v) for v in re.findall(r"> (5\.\d+)\)|<= (5\.\d+)\)", g2o) for v in v if v} == {5.991}
    assert "sqrt(5.991)" in g2o and re.search(r"sqrt\(5\.991\)", fn)


def test_the_forks_own_extractor_settings():
    """What this fork's launch files set (roslaunch/*.yaml, Examples/*/*.yaml): the tuple most of them share -- 1200 features, scale 1.1,
    6 levels, FAST 24 / 7 -- is the one tests/test_gpu_extract.py and bench.py's `fork_settings` leg run; prisms (nElType 2) and the dead
    inverse path (bUseInverse 0) are what the FEM tests assume."""
    import collections
    import glob
    tuples, eltype, inverse = collections.Counter(), collections.Counter(), collections.Counter()
    for f in glob.glob(os.path.join(REF, "roslaunch", "*.yaml")) + glob.glob(os.path.join(REF, "Examples", "*", "*.yaml")):
        t = open(f, errors="replace").read()
        g = {k: re.search(r"^ORBextractor\." + k + r":\s*([0-9.]+)", t, re.M) for k in ("nFeatures", "scaleFactor", "nLevels", "iniThFAST", "minThFAST")}
        if all(g.values()):
            tuples[tuple(float(v.group(1)) for v in g.values())] += 1
        m = re.search(r"^RelocParam\.nElType:\s*(\d+)", t, re.M)
        if m: eltype[int(m.group(1))] += 1
        m = re.search(r"^RelocParam\.bUseInverse:\s*(\d+)", t, re.M)
        if m: inverse[int(m.group(1))] += 1
    assert tuples.most_common(1)[0][0] == (1200.0, 1.1, 6.0, 24.0, 7.0) and tuples.most_common(1)[0][1] >= 10
    assert eltype.most_common(1)[0][0] == 2 and set(inverse) == {0}
    src = _ours("tests/test_gpu_extract.py") + _ours("bench.py")
    assert "(1200, 1.1, 6, 24, 7)" in src and "ORBextractor(1200, 1.1, 6, 24, 7)" in src
