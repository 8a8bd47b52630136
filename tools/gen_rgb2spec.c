/*
 * gen_rgb2spec.c — regenerate the sRGB -> sigmoid-polynomial coefficient table.
 *
 * The reference ships this table as a binary blob (src/spectral/srgb_spectrum_table.dat) that is
 * missing from the reference checkout (.MISSING_LARGE_BLOBS:3-4).  When the blob is absent the
 * reference regenerates it with spectral/rgb2spec_gen.jl (rgb2spec.jl:418-429); this program is a
 * restatement of that generator (Float64 Gauss-Newton in CIELAB, 3/8 Simpson over 283 wavelengths,
 * res 64), written from the algorithm description there:
 *
 *   init_tables          rgb2spec_gen.jl:169-214
 *   eval_residual!       rgb2spec_gen.jl:223-249
 *   eval_jacobian!       rgb2spec_gen.jl:252-272
 *   gauss_newton!        rgb2spec_gen.jl:275-305
 *   generate table loop  rgb2spec_gen.jl:332-412
 *
 * Output layout = load_srgb_table_binary (rgb2spec.jl:402-412):
 *   int32 res; float32 scale[res]; float32 coeffs[3][res][res][res][3] in Julia column-major order
 *   for dims [maxc, z, y, x, coeff]  (maxc fastest).
 *
 * Build: gcc -O2 -fopenmp -o gen_rgb2spec gen_rgb2spec.c -lm ; run: ./gen_rgb2spec out.dat [res]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CIE_LAMBDA_MIN 360.0
#define CIE_LAMBDA_MAX 830.0
#define CIE_SAMPLES 95
#define N_FINE ((CIE_SAMPLES - 1) * 3 + 1)
#define EPSILON 1e-4

/* CIE 1931 observer, 5 nm, 360..830 nm (public CIE data, same sampling as rgb2spec_gen.jl:20-88). */
static const double cie_x[CIE_SAMPLES] = {
    0.000129900000, 0.000232100000, 0.000414900000, 0.000741600000, 0.001368000000, 0.002236000000,
    0.004243000000, 0.007650000000, 0.014310000000, 0.023190000000, 0.043510000000, 0.077630000000,
    0.134380000000, 0.214770000000, 0.283900000000, 0.328500000000, 0.348280000000, 0.348060000000,
    0.336200000000, 0.318700000000, 0.290800000000, 0.251100000000, 0.195360000000, 0.142100000000,
    0.095640000000, 0.057950010000, 0.032010000000, 0.014700000000, 0.004900000000, 0.002400000000,
    0.009300000000, 0.029100000000, 0.063270000000, 0.109600000000, 0.165500000000, 0.225749900000,
    0.290400000000, 0.359700000000, 0.433449900000, 0.512050100000, 0.594500000000, 0.678400000000,
    0.762100000000, 0.842500000000, 0.916300000000, 0.978600000000, 1.026300000000, 1.056700000000,
    1.062200000000, 1.045600000000, 1.002600000000, 0.938400000000, 0.854449900000, 0.751400000000,
    0.642400000000, 0.541900000000, 0.447900000000, 0.360800000000, 0.283500000000, 0.218700000000,
    0.164900000000, 0.121200000000, 0.087400000000, 0.063600000000, 0.046770000000, 0.032900000000,
    0.022700000000, 0.015840000000, 0.011359160000, 0.008110916000, 0.005790346000, 0.004109457000,
    0.002899327000, 0.002049190000, 0.001439971000, 0.000999949300, 0.000690078600, 0.000476021300,
    0.000332301100, 0.000234826100, 0.000166150500, 0.000117413000, 0.000083075270, 0.000058706520,
    0.000041509940, 0.000029353260, 0.000020673830, 0.000014559770, 0.000010253980, 0.000007221456,
    0.000005085868, 0.000003581652, 0.000002522525, 0.000001776509, 0.000001251141};
static const double cie_y[CIE_SAMPLES] = {
    0.000003917000, 0.000006965000, 0.000012390000, 0.000022020000, 0.000039000000, 0.000064000000,
    0.000120000000, 0.000217000000, 0.000396000000, 0.000640000000, 0.001210000000, 0.002180000000,
    0.004000000000, 0.007300000000, 0.011600000000, 0.016840000000, 0.023000000000, 0.029800000000,
    0.038000000000, 0.048000000000, 0.060000000000, 0.073900000000, 0.090980000000, 0.112600000000,
    0.139020000000, 0.169300000000, 0.208020000000, 0.258600000000, 0.323000000000, 0.407300000000,
    0.503000000000, 0.608200000000, 0.710000000000, 0.793200000000, 0.862000000000, 0.914850100000,
    0.954000000000, 0.980300000000, 0.994950100000, 1.000000000000, 0.995000000000, 0.978600000000,
    0.952000000000, 0.915400000000, 0.870000000000, 0.816300000000, 0.757000000000, 0.694900000000,
    0.631000000000, 0.566800000000, 0.503000000000, 0.441200000000, 0.381000000000, 0.321000000000,
    0.265000000000, 0.217000000000, 0.175000000000, 0.138200000000, 0.107000000000, 0.081600000000,
    0.061000000000, 0.044580000000, 0.032000000000, 0.023200000000, 0.017000000000, 0.011920000000,
    0.008210000000, 0.005723000000, 0.004102000000, 0.002929000000, 0.002091000000, 0.001484000000,
    0.001047000000, 0.000740000000, 0.000520000000, 0.000361100000, 0.000249200000, 0.000171900000,
    0.000120000000, 0.000084800000, 0.000060000000, 0.000042400000, 0.000030000000, 0.000021200000,
    0.000014990000, 0.000010600000, 0.000007465700, 0.000005257800, 0.000003702900, 0.000002607800,
    0.000001836600, 0.000001293400, 0.000000910930, 0.000000641530, 0.000000451810};
static const double cie_z[CIE_SAMPLES] = {
    0.000606100000, 0.001086000000, 0.001946000000, 0.003486000000, 0.006450001000, 0.010549990000,
    0.020050010000, 0.036210000000, 0.067850010000, 0.110200000000, 0.207400000000, 0.371300000000,
    0.645600000000, 1.039050100000, 1.385600000000, 1.622960000000, 1.747060000000, 1.782600000000,
    1.772110000000, 1.744100000000, 1.669200000000, 1.528100000000, 1.287640000000, 1.041900000000,
    0.812950100000, 0.616200000000, 0.465180000000, 0.353300000000, 0.272000000000, 0.212300000000,
    0.158200000000, 0.111700000000, 0.078249990000, 0.057250010000, 0.042160000000, 0.029840000000,
    0.020300000000, 0.013400000000, 0.008749999000, 0.005749999000, 0.003900000000, 0.002749999000,
    0.002100000000, 0.001800000000, 0.001650001000, 0.001400000000, 0.001100000000, 0.001000000000,
    0.000800000000, 0.000600000000, 0.000340000000, 0.000240000000, 0.000190000000, 0.000100000000,
    0.000049999990, 0.000030000000, 0.000020000000, 0.000010000000, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define CIE_D65_NORM 10566.864005283874576
static const double cie_d65_raw[CIE_SAMPLES] = {
    46.6383, 49.3637, 52.0891, 51.0323, 49.9755, 52.3118, 54.6482, 68.7015, 82.7549, 87.1204, 91.486,
    92.4589, 93.4318, 90.057,  86.6823, 95.7736, 104.865, 110.936, 117.008, 117.41,  117.812, 116.336,
    114.861, 115.392, 115.923, 112.367, 108.811, 109.082, 109.354, 108.578, 107.802, 106.296, 104.79,
    106.239, 107.689, 106.047, 104.405, 104.225, 104.046, 102.023, 100.0,   98.1671, 96.3342, 96.0611,
    95.788,  92.2368, 88.6856, 89.3459, 90.0062, 89.8026, 89.5991, 88.6489, 87.6987, 85.4936, 83.2886,
    83.4939, 83.6992, 81.863,  80.0268, 80.1207, 80.2146, 81.2462, 82.2778, 80.281,  78.2842, 74.0027,
    69.7213, 70.6652, 71.6091, 72.979,  74.349,  67.9765, 61.604,  65.7448, 69.8856, 72.4863, 75.087,
    69.3398, 63.5927, 55.0054, 46.4182, 56.6118, 66.8054, 65.0941, 63.3828, 63.8434, 64.304,  61.8779,
    59.4519, 55.7054, 51.959,  54.6998, 57.4406, 58.8765, 60.3125};

static const double XYZ_TO_SRGB[3][3] = {{3.240479, -1.537150, -0.498535},
                                         {-0.969256, 1.875991, 0.041556},
                                         {0.055648, -0.204043, 1.057311}};
static const double SRGB_TO_XYZ[3][3] = {{0.412453, 0.357580, 0.180423},
                                         {0.212671, 0.715160, 0.072169},
                                         {0.019334, 0.119193, 0.950227}};

static double g_lambda[N_FINE];
static double g_rgbw[3][N_FINE];
static double g_white[3];

static double cie_interp(const double* data, double lambda, double post) {
    double x = lambda - CIE_LAMBDA_MIN;
    x *= (CIE_SAMPLES - 1) / (CIE_LAMBDA_MAX - CIE_LAMBDA_MIN);
    int offset = (int)floor(x);
    if (offset < 0) offset = 0;
    if (offset > CIE_SAMPLES - 2) offset = CIE_SAMPLES - 2;
    double weight = x - offset;
    return (1.0 - weight) * (data[offset] * post) + weight * (data[offset + 1] * post);
}

static void init_tables(void) {
    const double h = (CIE_LAMBDA_MAX - CIE_LAMBDA_MIN) / (N_FINE - 1);
    g_white[0] = g_white[1] = g_white[2] = 0.0;
    for (int i = 1; i <= N_FINE; ++i) {
        double lam = CIE_LAMBDA_MIN + (i - 1) * h;
        g_lambda[i - 1] = lam;
        double xyz[3] = {cie_interp(cie_x, lam, 1.0), cie_interp(cie_y, lam, 1.0), cie_interp(cie_z, lam, 1.0)};
        /* the reference divides the D65 table by its norm elementwise first, then interpolates */
        double d65n[2];
        (void)d65n;
        double I;
        {
            double x = lam - CIE_LAMBDA_MIN;
            x *= (CIE_SAMPLES - 1) / (CIE_LAMBDA_MAX - CIE_LAMBDA_MIN);
            int offset = (int)floor(x);
            if (offset < 0) offset = 0;
            if (offset > CIE_SAMPLES - 2) offset = CIE_SAMPLES - 2;
            double weight = x - offset;
            I = (1.0 - weight) * (cie_d65_raw[offset] / CIE_D65_NORM) + weight * (cie_d65_raw[offset + 1] / CIE_D65_NORM);
        }
        double weight = 3.0 / 8.0 * h;
        if (i == 1 || i == N_FINE) {
        } else if ((i - 2) % 3 == 2) {
            weight *= 2.0;
        } else {
            weight *= 3.0;
        }
        for (int k = 0; k < 3; ++k) {
            double acc = 0.0;
            for (int j = 0; j < 3; ++j) acc += XYZ_TO_SRGB[k][j] * xyz[j] * I * weight;
            g_rgbw[k][i - 1] = acc;
        }
        for (int j = 0; j < 3; ++j) g_white[j] += xyz[j] * I * weight;
    }
}

static double sigmoid(double x) { return 0.5 * x / sqrt(1.0 + x * x) + 0.5; }
static double smoothstep(double x) { return x * x * (3.0 - 2.0 * x); }

static double lab_f(double t) {
    const double d = 6.0 / 29.0;
    return t > d * d * d ? cbrt(t) : t / (3.0 * (d * d)) + 4.0 / 29.0;
}

static void rgb_to_lab(const double rgb[3], double lab[3]) {
    double xyz[3];
    for (int i = 0; i < 3; ++i) xyz[i] = SRGB_TO_XYZ[i][0] * rgb[0] + SRGB_TO_XYZ[i][1] * rgb[1] + SRGB_TO_XYZ[i][2] * rgb[2];
    double fx = lab_f(xyz[0] / g_white[0]), fy = lab_f(xyz[1] / g_white[1]), fz = lab_f(xyz[2] / g_white[2]);
    lab[0] = 116.0 * fy - 16.0;
    lab[1] = 500.0 * (fx - fy);
    lab[2] = 200.0 * (fy - fz);
}

static void eval_residual(double residual[3], const double coeffs[3], const double target[3]) {
    double out[3] = {0, 0, 0};
    for (int i = 0; i < N_FINE; ++i) {
        double ln = (g_lambda[i] - CIE_LAMBDA_MIN) / (CIE_LAMBDA_MAX - CIE_LAMBDA_MIN);
        double x = coeffs[0] * ln * ln + coeffs[1] * ln + coeffs[2];
        double s = sigmoid(x);
        for (int j = 0; j < 3; ++j) out[j] += g_rgbw[j][i] * s;
    }
    double ol[3], tl[3];
    rgb_to_lab(out, ol);
    rgb_to_lab(target, tl);
    for (int j = 0; j < 3; ++j) residual[j] = tl[j] - ol[j];
}

static void eval_jacobian(double jac[3][3], const double coeffs[3], const double target[3]) {
    double r0[3], r1[3], tmp[3];
    for (int i = 0; i < 3; ++i) {
        memcpy(tmp, coeffs, sizeof tmp);
        tmp[i] -= EPSILON;
        eval_residual(r0, tmp, target);
        memcpy(tmp, coeffs, sizeof tmp);
        tmp[i] += EPSILON;
        eval_residual(r1, tmp, target);
        for (int j = 0; j < 3; ++j) jac[j][i] = (r1[j] - r0[j]) / (2 * EPSILON);
    }
}

/* LU with partial pivoting (what Julia's `\` does for a dense square matrix); returns 0 if singular. */
static int solve3(double A[3][3], const double b[3], double x[3]) {
    double M[3][4];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) M[i][j] = A[i][j];
        M[i][3] = b[i];
    }
    for (int c = 0; c < 3; ++c) {
        int p = c;
        double best = fabs(M[c][c]);
        for (int r = c + 1; r < 3; ++r)
            if (fabs(M[r][c]) > best) {
                best = fabs(M[r][c]);
                p = r;
            }
        if (best == 0.0 || !isfinite(best)) return 0;
        if (p != c)
            for (int j = 0; j < 4; ++j) {
                double t = M[c][j];
                M[c][j] = M[p][j];
                M[p][j] = t;
            }
        for (int r = c + 1; r < 3; ++r) {
            double f = M[r][c] / M[c][c];
            for (int j = c; j < 4; ++j) M[r][j] -= f * M[c][j];
        }
    }
    for (int i = 2; i >= 0; --i) {
        double s = M[i][3];
        for (int j = i + 1; j < 3; ++j) s -= M[i][j] * x[j];
        x[i] = s / M[i][i];
    }
    return 1;
}

static void gauss_newton(double coeffs[3], const double target[3]) {
    double residual[3], jac[3][3], x[3];
    for (int it = 0; it < 15; ++it) {
        eval_residual(residual, coeffs, target);
        eval_jacobian(jac, coeffs, target);
        if (!solve3(jac, residual, x)) break;
        double maxc = 0.0;
        for (int i = 0; i < 3; ++i) {
            coeffs[i] -= x[i];
            if (fabs(coeffs[i]) > maxc) maxc = fabs(coeffs[i]);
        }
        if (maxc > 200.0)
            for (int i = 0; i < 3; ++i) coeffs[i] *= 200.0 / maxc;
        double r = residual[0] * residual[0] + residual[1] * residual[1] + residual[2] * residual[2];
        if (r < 1e-6) break;
    }
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s out.dat [res]\n", argv[0]);
        return 2;
    }
    int res = argc > 2 ? atoi(argv[2]) : 64;
    init_tables();
    float* scale = (float*)malloc(sizeof(float) * res);
    for (int k = 0; k < res; ++k) scale[k] = (float)smoothstep(smoothstep((double)k / (res - 1)));
    size_t n = (size_t)3 * res * res * res * 3;
    float* out = (float*)calloc(n, sizeof(float));
    /* Julia column-major index for coeffs[l,k,j,i,c] (1-based) with dims (3,res,res,res,3) */
#define IDX(l, k, j, i, c) ((size_t)(l) + 3 * ((size_t)(k) + res * ((size_t)(j) + res * ((size_t)(i) + (size_t)res * (c)))))
    for (int l = 0; l < 3; ++l) {
#pragma omp parallel for schedule(dynamic, 1)
        for (int j = 0; j < res; ++j) {
            double y = (double)j / (res - 1);
            for (int i = 0; i < res; ++i) {
                double x = (double)i / (res - 1);
                int start_k = res / 5;
                double opt[3], rgb[3];
                const double c0 = 360.0, c1 = 1.0 / (830.0 - 360.0);
                for (int pass = 0; pass < 2; ++pass) {
                    opt[0] = opt[1] = opt[2] = 0.0;
                    /* forward: k = start_k+1..res (1-based); backward: k = start_k+1 down to 1 */
                    int k = start_k + 1;
                    while (pass == 0 ? k <= res : k >= 1) {
                        double b = (double)scale[k - 1];
                        rgb[l] = b;
                        rgb[(l + 1) % 3] = x * b;
                        rgb[(l + 2) % 3] = y * b;
                        gauss_newton(opt, rgb);
                        double A = opt[0], B = opt[1], C = opt[2];
                        out[IDX(l, k - 1, j, i, 0)] = (float)(A * (c1 * c1));
                        out[IDX(l, k - 1, j, i, 1)] = (float)(B * c1 - 2 * A * c0 * (c1 * c1));
                        out[IDX(l, k - 1, j, i, 2)] = (float)(C - B * c0 * c1 + A * ((c0 * c1) * (c0 * c1)));
                        k += pass == 0 ? 1 : -1;
                    }
                }
            }
        }
        fprintf(stderr, "max component %d/3 done\n", l + 1);
    }
    FILE* f = fopen(argv[1], "wb");
    if (!f) {
        perror("fopen");
        return 1;
    }
    int32_t r32 = res;
    fwrite(&r32, 4, 1, f);
    fwrite(scale, 4, res, f);
    fwrite(out, 4, n, f);
    fclose(f);
    return 0;
}
