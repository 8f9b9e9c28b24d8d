/* A non-Python consumer of the C ABI: compiled by tests/test_abi_cpu.py with plain gcc against include/fbhip.h and
 * linked to libfbhip.so.  It touches only the host-side entry points (no GPU needed): version, layout queries, the
 * struct_size guard, context create / destroy and the "not bound" error path. */
#include <stdio.h>
#include <string.h>
#include "fbhip.h"

static fbhip_dims walker(void) {
    fbhip_dims d;
    memset(&d, 0, sizeof d);
    d.struct_size = sizeof d;
    d.batch = 1024; d.obs_dim = 24; d.action_dim = 6; d.goal_dim = 24; d.z_dim = 50;
    d.hidden_dim = 1024; d.feature_dim = 512; d.backward_hidden_dim = 526;
    d.preprocess = 1; d.norm_z = 1;
    return d;
}

int main(void) {
    if (fbhip_abi_version() != FBHIP_ABI_VERSION) { printf("FAIL abi %d != %d\n", fbhip_abi_version(), FBHIP_ABI_VERSION); return 1; }
    fbhip_dims d = walker();
    long long nf = fbhip_net_param_count(&d, FBHIP_NET_FORWARD), nb = fbhip_net_param_count(&d, FBHIP_NET_BACKWARD),
              na = fbhip_net_param_count(&d, FBHIP_NET_ACTOR);
    printf("params %lld %lld %lld\n", nf, nb, na);
    if (nf != 3363940 || nb != 317754 || na != 2211846) { printf("FAIL counts\n"); return 1; }
    if (fbhip_net_numel(&d, FBHIP_NET_FORWARD) < nf) { printf("FAIL numel\n"); return 1; }
    fbhip_tensor_desc t;
    if (fbhip_layout_entry(&d, FBHIP_NET_ACTOR, 0, &t) != FBHIP_OK) { printf("FAIL layout: %s\n", fbhip_last_error(NULL)); return 1; }
    printf("actor[0] %s %d x %d ld %d off %lld\n", t.name, t.rows, t.cols, t.ld, (long long)t.offset);
    if (strcmp(t.name, "obs_net.0.weight") != 0 || t.rows != 1024 || t.cols != 24) { printf("FAIL entry\n"); return 1; }
    if (fbhip_workspace_bytes(&d) == 0) { printf("FAIL workspace\n"); return 1; }
    /* every layout of every configuration flag: names, bounds, no overlap (also what the ASan build of the library runs) */
    for (int variant = 0; variant < 7; ++variant) {
        fbhip_dims v = walker();
        v.batch = 64; v.hidden_dim = 64; v.feature_dim = 32; v.backward_hidden_dim = 22; v.z_dim = 10;
        if (variant == 1) v.add_trunk = 1;
        if (variant == 2) v.preprocess = 0;
        if (variant == 3) v.boltzmann = 1;
        if (variant == 4) { v.preprocess = 0; v.discrete = 1; }
        if (variant == 5) v.sf = 1;
        if (variant == 6) { v.sf = 2; v.use_goal = 1; v.goal_dim = 3; }
        for (int net = 0; net < 3; ++net) {
            long long numel = fbhip_net_numel(&v, net), last_end = 0;
            int n = fbhip_layout_count(&v, net);
            if (numel < 0 || n < 0) { printf("FAIL variant %d net %d: %s\n", variant, net, fbhip_last_error(NULL)); return 1; }
            for (int i = 0; i < n; ++i) {
                if (fbhip_layout_entry(&v, net, i, &t) != FBHIP_OK || t.ld < t.cols || t.offset < 0 ||
                    t.offset + (long long)t.rows * t.ld > numel) { printf("FAIL entry %d of variant %d net %d\n", i, variant, net); return 1; }
                if (t.offset + (long long)t.rows * t.ld > last_end) last_end = t.offset + (long long)t.rows * t.ld;
            }
            if (fbhip_layout_entry(&v, net, n, &t) == FBHIP_OK) { printf("FAIL out-of-range entry accepted\n"); return 1; }
        }
        fbhip_ctx* vc = NULL;
        if (fbhip_workspace_bytes(&v) == 0 || fbhip_create(&v, &vc) != FBHIP_OK || fbhip_destroy(vc) != FBHIP_OK) { printf("FAIL create variant %d: %s\n", variant, fbhip_last_error(NULL)); return 1; }
    }

    /* a caller compiled against an older header (one field short) is refused, not read past its end */
    fbhip_dims shorter = d;
    shorter.struct_size = sizeof d - sizeof(int32_t);
    fbhip_ctx* ctx = NULL;
    if (fbhip_create(&shorter, &ctx) != FBHIP_E_INVALID || !strstr(fbhip_last_error(NULL), "struct_size")) { printf("FAIL struct_size guard\n"); return 1; }

    if (fbhip_create(&d, &ctx) != FBHIP_OK) { printf("FAIL create: %s\n", fbhip_last_error(NULL)); return 1; }
    fbhip_hparams hp;
    memset(&hp, 0, sizeof hp);
    hp.struct_size = sizeof hp;
    int rc = fbhip_update(ctx, &hp, NULL, FBHIP_PHASE_ALL, 0, NULL);
    if (rc != FBHIP_E_STATE) { printf("FAIL update on an unbound context returned %d\n", rc); return 1; }
    printf("unbound: %s\n", fbhip_last_error(ctx));
    if (fbhip_destroy(ctx) != FBHIP_OK) { printf("FAIL destroy\n"); return 1; }
    printf("OK\n");
    return 0;
}
