// The library's tuning switches in ONE place (include/vaeseg.h: vs_config).  Until round 6 they were ~30 getenv() calls cached in function-local statics
// all over the launchers: hidden process-global state nobody could inspect or change after the first call.  Now: one process-wide struct, filled on first
// use from the same environment variables (so the A/B scripts of tools/ keep working), read by every launcher through vs_cfg() on EVERY call, replaceable as
// a whole through vs_set_config() — the only writer.  Nothing here touches a stream or the device.
#include <mutex>
#include <stdlib.h>
#include "common.h"

static vs_config g_cfg;
static std::once_flag g_cfg_once;
static std::mutex g_cfg_mutex;

static int env_int(const char* name, int dflt) { const char* s = getenv(name); return s ? atoi(s) : dflt; }
static long long env_ll(const char* name, long long dflt) { const char* s = getenv(name); return s ? atoll(s) : dflt; }

extern "C" int vs_config_from_env(vs_config* c) {
    if (!c) return VS_EINVAL;
    c->k3_small = env_int("VS_K3_SMALL", 1);
    c->k3_tall = env_int("VS_K3_TALL", -1);
    c->k3_wgs_per_cu = env_int("VS_K3_WGS_PER_CU", 0);
    c->k3t_wgs_per_cu = env_int("VS_K3T_WGS_PER_CU", 2);
    c->k3f_min_wgs = env_int("VS_K3F_MIN_WGS", 512);
    c->mt_min_wgs = env_int("VS_MT_MIN_WGS", 1024);
    c->f32_limbs = env_int("VS_F32_LIMBS", 1);
    c->g1_limbs = env_int("VS_G1_LIMBS", 1);
    c->k3x_ck = env_int("VS_K3X_CK", 8);
    c->k3x_toeplitz = env_int("VS_K3X_TOEPLITZ", 1);
    c->fuse_wgrad = env_int("VS_FUSE_WGRAD", 1);
    c->epilogue_apply = env_int("VS_EPILOGUE_APPLY", 1);
    c->chain = env_int("VS_CHAIN", 1);
    c->k2s2_stream = env_int("VS_K2S2_STREAM", 1);
    c->k2s8_wgs_per_cu = env_int("VS_K2S8_WGS_PER_CU", 4);
    c->up_wgs_per_cu = env_int("VS_UP_WGS_PER_CU", 2);
    c->up_rb = env_int("VS_UP_RB", 0);
    c->wgrad_uber = env_int("VS_WGRAD_UBER", 1);
    c->wgrad_mpack = env_int("VS_WGRAD_MPACK", 1);
    c->wgrad_swap = env_int("VS_WGRAD_SWAP", 1);
    c->wgrad_big = env_int("VS_WGRAD_BIG", 1);
    c->wgrad_xcd = env_int("VS_WGRAD_XCD", 2);
    c->k3_short_tiles = env_int("VS_K3_SHORT_TILES", 2);
    c->wgrad_bias_fold = env_int("VS_WGRAD_BIAS_FOLD", 1);
    c->wgrad_wgs = env_ll("VS_WGRAD_WGS", 512);
    c->wgrad_f32_tiles = env_ll("VS_WGRAD_F32_TILES", 8);
    c->wgrad_group_wgs = env_ll("VS_WGRAD_GROUP_WGS", 0);
    c->wgrad_big_min_voxels = env_ll("VS_WGRAD_BIG_MIN_VOXELS", 400000);
    return VS_OK;
}

const vs_config& vs_cfg() {
    std::call_once(g_cfg_once, [] { vs_config_from_env(&g_cfg); });
    return g_cfg;
}

extern "C" int vs_config_bytes(void) { return (int)sizeof(vs_config); }

extern "C" int vs_get_config(vs_config* out) {
    if (!out) return VS_EINVAL;
    const vs_config& c = vs_cfg();
    std::lock_guard<std::mutex> lock(g_cfg_mutex);
    *out = c;
    return VS_OK;
}

extern "C" int vs_set_config(const vs_config* in) {
    if (!in) return VS_EINVAL;
    if (in->k3x_ck != 8 && in->k3x_ck != 16) return VS_EINVAL;
    if (in->k3_wgs_per_cu < 0 || in->k3t_wgs_per_cu < 1 || in->k2s8_wgs_per_cu < 1 || in->up_wgs_per_cu < 1 || in->wgrad_wgs < 1 || in->wgrad_f32_tiles < 1 ||
        in->wgrad_group_wgs < 0 || in->wgrad_big_min_voxels < 0 || in->mt_min_wgs < 0 || in->k3f_min_wgs < 0) return VS_EINVAL;
    vs_cfg();                                            // the environment is read first, never after a set
    std::lock_guard<std::mutex> lock(g_cfg_mutex);
    g_cfg = *in;
    return VS_OK;
}
