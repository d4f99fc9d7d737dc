// TEST INFRASTRUCTURE — not part of the product.  A C-ABI shim over the REFERENCE's own sum-product decoder
// (/root/reference/lib/data/MNC: zb2x.cpp, bnd/bnd.cpp, ansi/{cmatrix,nrutil,r}.cpp — MacKay's "belief network
// decoder", compiled from where the sources lie by oracle/build_ref.sh) that decodes ONE block exactly as the pybind11
// entry point `zb2x` does (lib/data/MNC/MNC_py.cpp:110-183): defaults, K / N, alist file, priors = the given bit
// probabilities, z = 0, `bndecode`, read out x.  Used only to generate tests/golden/ldpc_decode.npz and to pin the
// restatement in oracle/fgnn_oracle.py.
#include <stdint.h>
#include <string.h>
#include "zb2x.h"

// bias: N_total = k + n probabilities of bit = 1.  Outputs: x [k + n] hard decisions, q1 [k + n] pseudo-posteriors,
// *loops = iterations run, returns the number of violated checks (0 = decoded), < 0 on setup errors.
extern "C" int ref_bnd_decode(const char* afile, const double* bias, int k, int n, int bndloops, uint8_t* x,
                              double* q1, int* loops) {
    bnd_control bndc;
    bnd_param bndp;
    zb2x_control c;
    zb2x_vectors vec;
    zb2x_all all;
    alist_matrix a;
    all.bndp = &bndp; all.bndc = &bndc; all.vec = &vec; all.a = &a; all.c = &c;
    bndp.a = &a;
    bnd_defaults(&bndp, &bndc);
    c_defaults(&c);
    all.c->K = k;
    all.c->N = n;
    strcpy(all.c->Afile, afile);
    all.c->xsourceonly = 1;
    all.c->zfromfile = 0;
    all.c->zfixed = 0;
    all.bndc->loops = bndloops;
    make_sense(&c, &all);
    make_space(&c, &vec);
    if (read_allocate_alist(&a, (char*)afile) < 0) return -1;
    if (check_alist_MN(&a, &vec) < 0) return -2;
    hook_zb2x_vec_to_bnd(&bndp, &vec);
    bnd_allocate(&bndp, &bndc);
    set_up_priors(&vec, &c);
    for (int i = 0; i < k + n; ++i) vec.bias[i + 1] = bias[i];
    const int viol = bndecode(&bndp, &bndc);
    for (int i = 0; i < k + n; ++i) { x[i] = vec.x[i + 1]; q1[i] = bndp.q1[i + 1]; }
    *loops = bndc.loop - 1;
    bnd_free(&bndp, &bndc);
    zb2x_free(&all);
    return viol;
}

extern "C" void ref_bnd_defaults(double* tinydiv, int* doclip, double* clip, int* dofudge, double* fudge) {
    bnd_control bndc;
    bnd_param bndp;
    bnd_defaults(&bndp, &bndc);
    *tinydiv = bndc.tinydiv; *doclip = bndc.doclip; *clip = bndc.clip; *dofudge = bndc.dofudge; *fudge = bndc.fudge;
}
