"""Generate the polynomial coefficients used by include/pgas_detmath.h.

Run once at development time (needs mpmath); the output is pasted into the header.
Coefficients are exact Taylor coefficients rounded to the nearest double:
  sinpi(f) = f*pi + f^3 * sum_k S[k] f^(2k),  S[k] = (-1)^(k+1) pi^(2k+3)/(2k+3)!
  cospi(f) = 1 + f^2 * sum_k C[k] f^(2k),     C[k] = (-1)^(k+1) pi^(2k+2)/(2k+2)!
  exp(r)   = sum_k r^k / k!
  log(1+f) = 2s + s*R(z), s=f/(2+f), z=s^2, R(z) = sum_{n>=1} 2 z^n/(2n+1)
"""
import mpmath as mp

mp.mp.prec = 400


def hexd(x):
    return float(x).hex()


def main():
    pi = mp.pi
    pi_hi = mp.mpf(float(pi))
    pi_lo = pi - pi_hi
    print("PI_HI", hexd(pi_hi), "PI_LO", hexd(pi_lo))
    print("// sinpi S[k], k=0..8")
    for k in range(9):
        n = 2 * k + 3
        c = (-1) ** (k + 1) * pi**n / mp.factorial(n)
        print(f"    {hexd(c)}, /* {mp.nstr(c, 20)} */")
    print("// cospi C[k], k=0..8")
    for k in range(9):
        n = 2 * k + 2
        c = (-1) ** (k + 1) * pi**n / mp.factorial(n)
        print(f"    {hexd(c)}, /* {mp.nstr(c, 20)} */")
    print("// exp E[k]=1/k!, k=0..13")
    for k in range(14):
        c = 1 / mp.factorial(k)
        print(f"    {hexd(c)}, /* {mp.nstr(c, 20)} */")
    print("// log L[n]=2/(2n+1), n=1..12")
    for n in range(1, 13):
        c = mp.mpf(2) / (2 * n + 1)
        print(f"    {hexd(c)}, /* {mp.nstr(c, 20)} */")
    ln2 = mp.log(2)
    # ln2_hi with 21 trailing zero bits so that k*ln2_hi is exact for |k| < 2^20
    import struct
    b = struct.unpack("<Q", struct.pack("<d", float(ln2)))[0]
    b_hi = b & ~((1 << 21) - 1)
    ln2_hi = mp.mpf(struct.unpack("<d", struct.pack("<Q", b_hi))[0])
    print("LN2_HI_T", hexd(ln2_hi), "LN2_LO_T", hexd(ln2 - ln2_hi))
    ln2_h = mp.mpf(float(ln2))
    print("LN2_HI", hexd(ln2_h), "LN2_LO", hexd(ln2 - ln2_h))
    print("LOG2E", hexd(1 / ln2))
    print("SQRT2", hexd(mp.sqrt(2)))
    print("LOG_2PI", hexd(mp.log(2 * pi)))


if __name__ == "__main__":
    main()
