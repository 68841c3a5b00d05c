// CPU model of the device's pow_pos(double, double): same operations with fma(); reference powl()
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
static double log_pos(double x) {
  int e; double m = frexp(x, &e);            // m in [0.5, 1)
  if (m < 0.70710678118654752) { m *= 2.0; e -= 1; }
  const double n = m - 1.0, d = m + 1.0;
  double r = 1.0 / d;                        // device: v_rcp_f64 + one Newton step (<= 2.3e-15), corrected below
  r = r * (1.0 + 2.3e-15);                   // model the seed's error
  double f = n * r;
  f = fma(fma(-d, f, n), r, f);              // one correction of the quotient
  const double f2 = f * f;
  double p = 1.0 / 21.0;
  p = fma(p, f2, 1.0 / 19.0); p = fma(p, f2, 1.0 / 17.0); p = fma(p, f2, 1.0 / 15.0); p = fma(p, f2, 1.0 / 13.0);
  p = fma(p, f2, 1.0 / 11.0); p = fma(p, f2, 1.0 / 9.0); p = fma(p, f2, 1.0 / 7.0); p = fma(p, f2, 1.0 / 5.0);
  p = fma(p, f2, 1.0 / 3.0);
  p = p * f2;
  const double tf = f + f, ed = (double)e;
  const double t = fma(ed, 0x1.a39ef35793c76p-33, tf * p);
  return fma(ed, 0x1.62e42fee00000p-1, tf + t);
}
static double exp_(double y) {
  const double k = rint(y * 1.4426950408889634);
  double r = fma(-k, 0x1.62e42fefa39efp-1, y);
  r = fma(-k, 0x1.abc9e3b39803fp-56, r);
  double p = 1.0 / 479001600.0;
  p = fma(p, r, 1.0 / 39916800.0); p = fma(p, r, 1.0 / 3628800.0); p = fma(p, r, 1.0 / 362880.0); p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0); p = fma(p, r, 1.0 / 720.0); p = fma(p, r, 1.0 / 120.0); p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0); p = fma(p, r, 0.5); p = fma(p, r, 1.0); p = fma(p, r, 1.0);
  return ldexp(p, (int)k);
}
int main(void) {
  double worst = 0, worst_log = 0, wx = 0, wb = 0, worst_old = 0;
  srand(7);
  for (int i = 0; i < 4000000; ++i) {
    const double u = rand() / (double)RAND_MAX, v = rand() / (double)RAND_MAX;
    const double x = exp((u * 2 - 1) * 9.2);         // 1e-4 .. 1e4
    const double b = 0.2 + 4.8 * v;
    const long double ref = powl((long double)x, (long double)b);
    const double got = exp_(b * log_pos(x));
    const double old = exp(b * log(x));
    const double err = fabs((double)((got - ref) / ref)), erro = fabs((double)((old - ref) / ref));
    const double el = fabs((double)((log_pos(x) - logl((long double)x)) / (fabsl(logl((long double)x)) + 1e-300L)));
    if (err > worst) { worst = err; wx = x; wb = b; }
    if (erro > worst_old) worst_old = erro;
    if (el > worst_log && fabs(x - 1) > 1e-3) worst_log = el;
  }
  printf("worst rel err new %.3e (x %.6g b %.4g)  libm exp(b log x) %.3e  log rel err %.3e\n", worst, wx, wb, worst_old, worst_log);
  // near 1
  double wn = 0;
  for (int i = -2000; i <= 2000; ++i) { double x = 1.0 + i * 1e-7; if (x == 1.0) continue; double e = fabs((double)((log_pos(x) - logl((long double)x)) / logl((long double)x))); if (e > wn) wn = e; }
  printf("log near 1: worst rel %.3e\n", wn);
  return 0;
}
