// Host build (g++) of tcdiff_amd/csrc/fk_math.h: the per-pose forward / reverse-mode functions the HIP kernels of
// csrc/train.hip call, exported for tests/test_train_cpu.py (compared there with torch autograd through the oracle).
#include "fk_math.h"

static FkSkel make_skel(const int* parents, const float* offsets) {
    FkSkel sk;
    for (int j = 0; j < TC_FK_J; ++j) sk.has_children[j] = 0;
    for (int j = 0; j < TC_FK_J; ++j) {
        sk.parent[j] = parents[j];
        if (parents[j] >= 0) sk.has_children[parents[j]] = 1;
        for (int k = 0; k < 3; ++k) sk.off[j][k] = offsets[3 * j + k];
    }
    return sk;
}

extern "C" {
void host_ax_from_6v(const float* d6, long n, float* aa) {
    for (long i = 0; i < n; ++i) {
        const V3 a = axis_angle_from_quat(quat_from_6d(d6 + 6 * i));
        aa[3 * i] = a.x; aa[3 * i + 1] = a.y; aa[3 * i + 2] = a.z;
    }
}
void host_ax_from_6v_bwd(const float* d6, const float* g_aa, long n, float* g6) {
    for (long i = 0; i < n; ++i) ax_from_6v_bwd(d6 + 6 * i, v3(g_aa[3 * i], g_aa[3 * i + 1], g_aa[3 * i + 2]), g6 + 6 * i);
}
void host_fk(const float* aa, const float* root, long n, const int* parents, const float* offsets, float* joints) {
    const FkSkel sk = make_skel(parents, offsets);
    for (long i = 0; i < n; ++i) fk_forward(aa + 72 * i, root + 3 * i, sk, joints + 72 * i, nullptr);
}
void host_fk_bwd(const float* aa, const float* g_joints, long n, const int* parents, const float* offsets, float* g_aa,
                 float* g_root) {
    const FkSkel sk = make_skel(parents, offsets);
    for (long i = 0; i < n; ++i) fk_backward(aa + 72 * i, sk, g_joints + 72 * i, g_aa + 72 * i, g_root + 3 * i);
}
}
