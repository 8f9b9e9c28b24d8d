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
