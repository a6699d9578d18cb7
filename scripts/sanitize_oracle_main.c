/* Driver for scripts/sanitize_cpu.sh: exercises every oracle entry point under ASan/UBSan. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void orc_interp1_scan_sorted(const double*, const double*, size_t, const double*, size_t, double, double*);
void orc_interp1_bracket(const double*, const double*, size_t, const double*, size_t, double, double*, int);
int orc_interp1_arma(const double*, const double*, size_t, const double*, size_t, double, double*);
void orc_interp1_uniform(double, double, const double*, size_t, const double*, size_t, double, double*, int);
void orc_interp2_bilinear(const double*, size_t, const double*, size_t, const double*, const double*, const double*, size_t, double, double*, int);
void orc_interp2_bilinear_uniform(double, double, size_t, double, double, size_t, const double*, const double*, const double*, size_t, double, double*, int);
void orc_restrict_f32(const float*, const uint16_t*, const float*, const uint16_t*, float, float, uint32_t, float*, size_t);
void orc_masked_mean_f32(const float*, const uint32_t*, size_t, size_t, int, float*, uint32_t*);
void orc_splitmix_uniform(uint64_t, double*, size_t);
typedef struct { float vth, a1, a2, b1, b2, I, L; double newton_tol; uint32_t newton_max_iter, n_spikes; float time_horizon;
                 uint32_t n_grid, n_real; float beta_mean, beta_stddev; uint64_t seed; int math_mode, mean_quirk; uint32_t max_events, real_offset; } P;
void orc_edm_default_params(P*);
int orc_edm_compute_f(const P*, const double*, double*, uint16_t*, float*, float*, float*, float*, uint16_t*, float*, uint16_t*,
                      uint32_t*, float*, double*, int);

int main(void)
{
    enum { NG = 777, NQ = 5001 };
    double *x = malloc(NG * 8), *y = malloc(NG * 8), *q = malloc(NQ * 8), *o = malloc(NQ * 8), *q2 = malloc(NQ * 8);
    orc_splitmix_uniform(1, x, NG);
    for (int i = 1; i < NG; ++i) x[i] += x[i - 1];
    for (int i = 0; i < NG; ++i) y[i] = sin(x[i]);
    orc_splitmix_uniform(2, q, NQ);
    for (int i = 0; i < NQ; ++i) q[i] = q[i] * (x[NG - 1] - x[0]) * 1.1 + x[0] - 0.01;
    q[0] = NAN; q[1] = x[0]; q[2] = x[NG - 1];
    if (orc_interp1_arma(x, y, NG, q, NQ, NAN, o)) return 1;
    orc_interp1_bracket(x, y, NG, q, NQ, -1.0, o, 2);
    orc_interp1_uniform(0.5, 0.01, y, NG, q, NQ, NAN, o, 2);
    orc_splitmix_uniform(3, q2, NQ);
    double xg[20], yg[15], z[300];
    for (int i = 0; i < 20; ++i) xg[i] = i * 0.05;
    for (int i = 0; i < 15; ++i) yg[i] = i * 0.07;
    for (int i = 0; i < 300; ++i) z[i] = cos(0.1 * i);
    for (int i = 0; i < NQ; ++i) { q[i] = q2[i] * 1.1 - 0.05; }
    orc_interp2_bilinear(xg, 20, yg, 15, z, q, q2, NQ, NAN, o, 2);
    orc_interp2_bilinear_uniform(0, 0.05, 20, 0, 0.07, 15, z, q, q2, NQ, NAN, o, 2);
    float t0[30], t1[30], out[30], mean[3];
    uint16_t i0[30], i1[30];
    uint32_t acc[10], cnt;
    for (int i = 0; i < 30; ++i) { t0[i] = 4.f + 0.01f * i; t1[i] = 5.5f; i0[i] = (uint16_t)(500 + i); i1[i] = (uint16_t)(501 + i); }
    for (int i = 0; i < 10; ++i) acc[i] = i % 3 != 0;
    orc_restrict_f32(t0, i0, t1, i1, 5.f, 3.f, 1024, out, 30);
    orc_masked_mean_f32(out, acc, 10, 3, 0, mean, &cnt);
    orc_masked_mean_f32(out, acc, 10, 3, 1, mean, &cnt);
    P p;
    orc_edm_default_params(&p);
    p.n_real = 2; p.n_grid = 512; p.beta_stddev = 0.2f;
    double Z[3] = {0.3310, 0.6914, 1.3557}, f[3], sums[7];   /* partial block: 2S+1 */
    uint16_t seed[8] = {0};
    if (orc_edm_compute_f(&p, Z, f, seed, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, sums, 2)) return 2;
    printf("oracle under sanitizers ok: f = %g %g %g\n", f[0], f[1], f[2]);
    free(x); free(y); free(q); free(o); free(q2);
    return 0;
}
