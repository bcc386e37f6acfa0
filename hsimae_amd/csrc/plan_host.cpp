// Host-only build of the planning code (plan.h) behind the library's own C entry points, for
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -shared -fPIC plan_host.cpp -o libhsimae_plan_asan.so
// (python -m hsimae_amd.build --asan).  Never loaded by the product; tests/test_plan_asan_cpu.py drives it.
#include "plan.h"

using namespace hsplan;

extern "C" {
int hsimae_param_layout(const hsimae_config* cfg, int64_t* offsets, int64_t* sizes, int max_entries) {
    return param_layout(cfg, offsets, sizes, max_entries);
}
int64_t hsimae_wpk_elems(const hsimae_config* cfg) { return wpk_elems(cfg); }
int64_t hsimae_pack_table_bytes(const hsimae_config* cfg) { return pack_table_bytes(cfg); }
int hsimae_build_pack_table(const hsimae_config* cfg, const float* params_dev, hs_bf16* wpk_dev, void* table_host) {
    return build_pack_table(cfg, params_dev, wpk_dev, table_host);
}
int64_t hsimae_workspace_bytes(const hsimae_config* cfg, int32_t N, int32_t len_t, int32_t len_l) {
    return workspace_bytes(cfg, N, len_t, len_l);
}
int32_t hsimae_wgrad_msplit(int32_t tiles, int64_t M) { return wgrad_msplit(tiles, M); }
int64_t hsimae_wgrad_slab_bytes(const hsimae_config* cfg, int32_t side_stream) { return wgrad_slab_bytes(cfg, side_stream); }
}
