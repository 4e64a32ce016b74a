// Host-only pieces of include/orbslam_hip.hpp on a toy map, driven by tests/test_fuse_and_projection.py: reads
//   nmp nkp nlist, then mp_obs[nmp] mp_bad[nmp] mp_in_kf[nmp] kf_mp[nkp], then list[] visible[] best[] idx[] (nlist each)
// from stdin and prints, for ORBmatcher::fuseReplay and ORBmatcher::fuseReplaySim3 (each on a fresh copy of the map):
//   nFused nops / the ops {kind a b} / the final mp_obs, mp_bad, mp_in_kf, kf_mp (Sim3: + vpReplacePoint[nlist]).
// No device call is made: these loops are the reference's loop tails (ORBmatcher.cc:1046-1053, 1149-1170, 1194-1205, 1279-1296).
#include <cstdio>
#include <vector>

#include "orbslam_hip.hpp"

using namespace orbslam_hip;

struct Toy {
    std::vector<int> obs, bad, in_kf, kf, list, vpReplace, already;
    std::vector<int> ops;
    int at(int i) const { return list[i]; }
    bool null(int h) const { return h < 0; }
    bool isBad(int h) const { return bad[h] != 0; }
    bool isInKeyFrame(int h) const { return in_kf[h] >= 0; }
    bool alreadyFound(int h) const { return already[h] != 0; }   // snapshot, filled before the loop
    int slotOwner(int idx) const { return kf[idx]; }
    int observations(int h) const { return obs[h]; }
    void replace_(int dead, int heir)
    {
        const int slot = in_kf[dead];
        bad[dead] = 1;
        if (slot >= 0) { in_kf[dead] = -1; kf[slot] = heir; in_kf[heir] = slot; obs[heir] += 1; ops.insert(ops.end(), {2, heir, dead}); }
        else ops.insert(ops.end(), {1, dead, heir});
    }
    void replace(int dead, int heir) { replace_(dead, heir); }
    void add(int h, int idx) { kf[idx] = h; in_kf[h] = idx; obs[h] += 1; ops.insert(ops.end(), {0, h, idx}); }
    void recordReplace(int i, int h) { vpReplace[i] = h; ops.insert(ops.end(), {3, i, h}); }
};

static void read_vec(std::vector<int> &v, int n)
{
    v.resize(n);
    for (int i = 0; i < n; ++i)
        if (scanf("%d", &v[i]) != 1) { fprintf(stderr, "short input\n"); exit(2); }
}
static void print_vec(const std::vector<int> &v)
{
    for (size_t i = 0; i < v.size(); ++i) printf("%d ", v[i]);
    printf("\n");
}

int main()
{
    int nmp, nkp, nlist;
    if (scanf("%d %d %d", &nmp, &nkp, &nlist) != 3) return 2;
    Toy base;
    read_vec(base.obs, nmp); read_vec(base.bad, nmp); read_vec(base.in_kf, nmp); read_vec(base.kf, nkp);
    std::vector<int> visible, best, idx;
    read_vec(base.list, nlist); read_vec(visible, nlist); read_vec(best, nlist); read_vec(idx, nlist);
    std::vector<orbm_projected_point> proj(nlist);
    for (int i = 0; i < nlist; ++i) { proj[i] = orbm_projected_point(); proj[i].visible = visible[i]; }
    std::vector<int32_t> b(best.begin(), best.end()), ix(idx.begin(), idx.end());
    {
        Toy t = base;
        const int n = ORBmatcher::fuseReplay(proj, b, ix, t);
        printf("%d %zu\n", n, t.ops.size() / 3);
        print_vec(t.ops); print_vec(t.obs); print_vec(t.bad); print_vec(t.in_kf); print_vec(t.kf);
    }
    {
        Toy t = base;
        t.vpReplace.assign(nlist, -1);
        t.already.assign(nmp, 0);
        for (int k = 0; k < nkp; ++k)                       // pKF->GetMapPoints(), KeyFrame.cc:274-287
            if (t.kf[k] >= 0 && !t.bad[t.kf[k]]) t.already[t.kf[k]] = 1;
        const int n = ORBmatcher::fuseReplaySim3(proj, b, ix, t);
        printf("%d %zu\n", n, t.ops.size() / 3);
        print_vec(t.ops); print_vec(t.obs); print_vec(t.bad); print_vec(t.in_kf); print_vec(t.kf); print_vec(t.vpReplace);
    }
    return 0;
}
