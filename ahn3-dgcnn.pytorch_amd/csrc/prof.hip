// prof.hip -- see prof.h
#include "prof.h"

#include "dgcnn_hip.h"

namespace dgcnn {
namespace {
constexpr int PROF_MAX = 256;  // samples kept per region
struct Region {
    int n = 0;
    bool open = false;
    hipEvent_t e0[PROF_MAX], e1[PROF_MAX];
};
struct ProfState {
    bool on = false, created = false;
    Region r[PROF_NREGION];
} g_prof;
const char *const kNames[PROF_NREGION] = {"knn_bound16s_kernel", "knn_filter16_kernel", "knn_refine_kernel",
                                          "edge_reduce_kernel", "edge_rgather_kernel", "edge_mlp_fwd_kernel",
                                          "edge_mlp_bwd_kernel", "head_conv6_kernel", "head_conv7_kernel", "head_conv8_kernel",
                                          "head_dz7_kernel", "head_dcat_kernel", "head_dw8_kernel", "head_dw7_kernel"};
}  // namespace

void prof_begin(int region, hipStream_t st)
{
    Region &r = g_prof.r[region];
    r.open = g_prof.on && r.n < PROF_MAX;
    if (r.open) (void)hipEventRecord(r.e0[r.n], st);
}
void prof_end(int region, hipStream_t st)
{
    Region &r = g_prof.r[region];
    if (r.open) (void)hipEventRecord(r.e1[r.n++], st);
    r.open = false;
}

}  // namespace dgcnn

using namespace dgcnn;

extern "C" int dgcnn_profile_enable(int on)
{
    if (on && !g_prof.created) {
        for (int g = 0; g < PROF_NREGION; ++g)
            for (int i = 0; i < PROF_MAX; ++i)
                // (no system-scope fence with the record: the events only time kernels of one stream; the default record
                // also writes the caches back, which costs the kernels around it several microseconds)
                if (hipEventCreateWithFlags(&g_prof.r[g].e0[i], hipEventDisableSystemFence) != hipSuccess ||
                    hipEventCreateWithFlags(&g_prof.r[g].e1[i], hipEventDisableSystemFence) != hipSuccess)
                    return 1;
        g_prof.created = true;
    }
    g_prof.on = on != 0;
    if (on)
        for (int g = 0; g < PROF_NREGION; ++g) g_prof.r[g].n = 0;  // enable(0) keeps the samples
    return 0;
}

extern "C" int dgcnn_profile_resume(void)
{
    if (!g_prof.created) return 1;
    g_prof.on = true;  // the samples recorded so far are kept
    return 0;
}

extern "C" int dgcnn_profile_regions(void) { return PROF_NREGION; }

extern "C" const char *dgcnn_profile_region_name(int region)
{
    return region >= 0 && region < PROF_NREGION ? kNames[region] : "";
}

extern "C" int dgcnn_profile_samples(int region)
{
    return region >= 0 && region < PROF_NREGION ? g_prof.r[region].n : 0;
}

extern "C" float dgcnn_profile_mean_ms(int region)
{
    if (region < 0 || region >= PROF_NREGION) return -1.0f;
    const Region &r = g_prof.r[region];
    if (r.n == 0) return 0.0f;
    float sum = 0.0f;
    for (int i = 0; i < r.n; ++i) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, r.e0[i], r.e1[i]) != hipSuccess) return -1.0f;
        sum += ms;
    }
    return sum / (float)r.n;
}
