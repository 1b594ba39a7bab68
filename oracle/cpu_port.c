/*
 * cpu_port.c -- plain-C (OpenMP) restatement of the KSD hot path, TEST INFRASTRUCTURE ONLY.
 *
 * Same algorithm as oracle/circuit.py and oracle/stein.py (which cite the reference lines they
 * follow): gate-by-gate complex128 statevector simulation in the gate order of
 * quantum_born_machine.py:57-128, one circuit per OpenMP thread; Stein Gram rows by the closed form
 * of stein_utils.py:138-197; y = K q.  It exists to (a) cross-check the
 * NumPy oracle with a third implementation and (b) give bench.py's `cpu_baseline` leg a
 * multi-threaded CPU number ("kind": "port").  Nothing in tensornetworks_amd/ links or loads it.
 *
 * Build: make -C oracle   ->  oracle/_build/libcpu_port.so
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { double re, im; } cplx;

static void apply_1q(cplx* s, int n, int w, const double U[8]) {
  const long stride = 1L << (n - 1 - w);
  const long N = 1L << n;
  for (long blk = 0; blk < N; blk += 2 * stride)
    for (long i = blk; i < blk + stride; ++i) {
      const cplx a = s[i], b = s[i + stride];
      s[i].re = U[0] * a.re - U[1] * a.im + U[2] * b.re - U[3] * b.im;
      s[i].im = U[0] * a.im + U[1] * a.re + U[2] * b.im + U[3] * b.re;
      s[i + stride].re = U[4] * a.re - U[5] * a.im + U[6] * b.re - U[7] * b.im;
      s[i + stride].im = U[4] * a.im + U[5] * a.re + U[6] * b.im + U[7] * b.re;
    }
}

static void gate_h(cplx* s, int n, int w) {
  const double h = 0.70710678118654752440;
  const double U[8] = {h, 0, h, 0, h, 0, -h, 0};
  apply_1q(s, n, w, U);
}
static void gate_rx(cplx* s, int n, int w, double t) {
  const double c = cos(t / 2), sn = sin(t / 2);
  const double U[8] = {c, 0, 0, -sn, 0, -sn, c, 0};
  apply_1q(s, n, w, U);
}
static void gate_ry(cplx* s, int n, int w, double t) {
  const double c = cos(t / 2), sn = sin(t / 2);
  const double U[8] = {c, 0, -sn, 0, sn, 0, c, 0};
  apply_1q(s, n, w, U);
}
static void gate_rz(cplx* s, int n, int w, double t) {
  const double c = cos(t / 2), sn = sin(t / 2);
  const double U[8] = {c, -sn, 0, 0, 0, 0, c, sn};
  apply_1q(s, n, w, U);
}
static void gate_cnot(cplx* s, int n, int c, int t) {
  const long cm = 1L << (n - 1 - c), tm = 1L << (n - 1 - t), N = 1L << n;
  for (long i = 0; i < N; ++i)
    if ((i & cm) && !(i & tm)) { const cplx a = s[i]; s[i] = s[i | tm]; s[i | tm] = a; }
}
static void gate_cz(cplx* s, int n, int a, int b) {
  const long am = 1L << (n - 1 - a), bm = 1L << (n - 1 - b), N = 1L << n;
  for (long i = 0; i < N; ++i)
    if ((i & am) && (i & bm)) { s[i].re = -s[i].re; s[i].im = -s[i].im; }
}

/* gate order: quantum_born_machine.py:58-87 (0), :90-111 (1), :114-128 (2) */
static void run_circuit(int ansatz, int n, int L, const double* th, cplx* s) {
  const long N = 1L << n;
  memset(s, 0, sizeof(cplx) * N);
  s[0].re = 1.0;
  int p = 0;
  if (ansatz == 0 || ansatz == 1) {
    for (int i = 0; i < n; ++i) gate_h(s, n, i);
    for (int l = 0; l < L; ++l) {
      for (int i = 0; i < n; ++i) { gate_rx(s, n, i, th[p++]); gate_ry(s, n, i, th[p++]); gate_rz(s, n, i, th[p++]); }
      if (n > 1) {
        if (ansatz == 0) {
          for (int i = 0; i + 1 < n; ++i) gate_cnot(s, n, i, i + 1);
          if (n > 2) gate_cnot(s, n, n - 1, 0);
          if (l % 2 == 0 && n > 2) for (int i = 0; i < n - 2; i += 2) gate_cz(s, n, i, i + 2);
        } else {
          for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) gate_cz(s, n, i, j);
        }
      }
    }
  } else {
    for (int l = 0; l < L; ++l) {
      for (int i = 0; i < n; ++i) { gate_ry(s, n, i, th[p++]); gate_rz(s, n, i, th[p++]); }
      if (n > 1) {
        for (int i = 0; i + 1 < n; ++i) gate_cnot(s, n, i, i + 1);
        if (n > 2) gate_cnot(s, n, n - 1, 0);
      }
    }
  }
}

/* thetas [B, P] -> probs [B, 2^n]; returns threads used */
int port_circuit_probs(int ansatz, int n, int L, int B, int P, const double* thetas, double* probs) {
  const long N = 1L << n;
  int used = 1;
#pragma omp parallel
  {
#ifdef _OPENMP
#pragma omp single
    used = omp_get_num_threads();
#endif
    cplx* s = (cplx*)malloc(sizeof(cplx) * N);
#pragma omp for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
      run_circuit(ansatz, n, L, thetas + (long)b * P, s);
      double* q = probs + (long)b * N;
      for (long i = 0; i < N; ++i) q[i] = s[i].re * s[i].re + s[i].im * s[i].im;
    }
    free(s);
  }
  return used;
}

/* rows (+p, -p) for p in [p_begin, p_end), optional base row first (parameter shift pi/2) */
int port_paramshift_probs(int ansatz, int n, int L, int P, const double* theta, int p_begin, int p_end,
                          int include_base, double* probs) {
  const int B = (include_base ? 1 : 0) + 2 * (p_end - p_begin);
  double* th = (double*)malloc(sizeof(double) * (size_t)B * (P > 0 ? P : 1));
  for (int b = 0; b < B; ++b) {
    memcpy(th + (long)b * P, theta, sizeof(double) * P);
    const int bb = b - (include_base ? 1 : 0);
    if (bb >= 0) th[(long)b * P + p_begin + bb / 2] += (bb & 1) ? -M_PI_2 : M_PI_2;
  }
  const int used = port_circuit_probs(ansatz, n, L, B, P, th, probs);
  free(th);
  return used;
}

/* K rows [r0, r1): closed form of k_p (see oracle/stein.py gram_closed_form) */
void port_gram_rows(int n, double length_scale, const double* S, long r0, long r1, double* K) {
  const long N = 1L << n;
  const double a = exp(-1.0 / (n * length_scale));
  const double c_same = 1.0 - a, c_diff = 1.0 - 1.0 / a;
  double apow[64];
  for (int d = 0; d <= n; ++d) apow[d] = exp(-(double)d / (n * length_scale));
#pragma omp parallel for schedule(static)
  for (long i = r0; i < r1; ++i)
    for (long j = 0; j < N; ++j) {
      const long x = i ^ j;
      double tot = 0.0;
      for (int b = 0; b < n; ++b) {
        const double si = S[i * n + b], sj = S[j * n + b];
        const double c = ((x >> (n - 1 - b)) & 1) ? c_diff : c_same;
        tot += si * sj - c * (si + sj) + 2.0 * c;
      }
      K[(i - r0) * N + j] = apow[__builtin_popcountl(x)] * tot;
    }
}

/* y[r] = sum_j K[r, j] q[j] for the given rows; returns sum_r q[r0 + r] y[r] */
double port_gemv_rows(int n, const double* K, long r0, long r1, const double* q, double* y) {
  const long N = 1L << n;
  double tot = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : tot)
  for (long r = 0; r < r1 - r0; ++r) {
    double acc = 0.0;
    const double* row = K + r * N;
    for (long j = 0; j < N; ++j) acc += row[j] * q[j];
    y[r] = acc;
    tot += q[r0 + r] * acc;
  }
  return tot;
}

/* v <- K_base v = M^{(x) n} v, M = [[1, a], [a, 1]]: n butterfly passes (oracle/stein.py kbase_apply) */
static void kbase_inplace(double* v, int n, double a) {
  const long N = 1L << n;
  for (int b = 0; b < n; ++b) {
    const long st = 1L << b;
#pragma omp parallel for schedule(static)
    for (long p = 0; p < N / 2; ++p) {
      const long i = ((p & ~(st - 1)) << 1) | (p & (st - 1));
      const double x0 = v[i], x1 = v[i + st];
      v[i] = x0 + a * x1;
      v[i + st] = a * x0 + x1;
    }
  }
}

/* Matrix-free y = K_p q (oracle/stein.py stein_matvec_kron term by term): the CPU side of
 * bench.py's baseline where the dense Gram does not fit (n = 20).  Returns q . y.  work: 2 * 2^n doubles. */
double port_kron_matvec(int n, double length_scale, const double* S, const double* q, double* y, double* work) {
  const long N = 1L << n;
  const double a = exp(-1.0 / (n * length_scale));
  double* u = work;
  double* w = work + N;
#pragma omp parallel for schedule(static)
  for (long i = 0; i < N; ++i) { u[i] = q[i]; y[i] = 0.0; }
  kbase_inplace(u, n, a);
  for (int b = 0; b < n; ++b) {
    const long fm = 1L << (n - 1 - b);               /* tuple position b = bit n-1-b of the index */
#pragma omp parallel for schedule(static)
    for (long i = 0; i < N; ++i) w[i] = S[i * n + b] * q[i];
    kbase_inplace(w, n, a);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < N; ++i) {
      const double sb = S[i * n + b];
      const double du = u[i] - u[i ^ fm], dw = w[i] - w[i ^ fm];
      y[i] += sb * w[i] - sb * du - dw + 2.0 * du;
    }
  }
  double tot = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : tot)
  for (long i = 0; i < N; ++i) tot += q[i] * y[i];
  return tot;
}

int port_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
