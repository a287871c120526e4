"""Goldilocks field, quadratic extension and the NTT/LDE definition.  TEST INFRASTRUCTURE.

Follows gnark-plonky2-verifier/goldilocks/base.go:33-42 (p, generator 7, 2-adicity 32,
2^32-th root 1753635133440165772) and goldilocks/quadratic_extension.go:9-10,59-73
(F[x]/(x^2 - 7)).  The transform itself lives in the un-vendored plonky2 fork
(wormhole-foundation/plonky2-near @ 2244a9d, Cargo.toml:44-47): restated from its
published definition -- values[k] = sum_j coeffs[j] * w^(j k), w = primitive n-th root
= POWER_OF_TWO_GENERATOR^(2^32/n); coset LDE = evaluate the zero-padded polynomial on
shift * <w_N>.  PARITY UNPINNED for the NTT (the reference holds no FFT vector): only
self-consistency (inverse, evaluation at a point) is checked.
"""
P = 2**64 - 2**32 + 1
GENERATOR = 7
TWO_ADICITY = 32
POWER_OF_TWO_GENERATOR = 1753635133440165772
W = 7  # extension non-residue
assert pow(POWER_OF_TWO_GENERATOR, 1 << 32, P) == 1 and pow(POWER_OF_TWO_GENERATOR, 1 << 31, P) == P - 1


def root_of_unity(log_n):
    return pow(POWER_OF_TWO_GENERATOR, 1 << (TWO_ADICITY - log_n), P)


def inv(a):
    return pow(a, P - 2, P)


def bitrev(i, bits):
    return int(format(i, "0%db" % bits)[::-1], 2) if bits else 0


def ntt(a, inverse=False):
    """natural order in -> natural order out (O(n log n), Python ints)"""
    n = len(a)
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    a = [a[bitrev(i, log_n)] for i in range(n)]
    w_n = root_of_unity(log_n)
    if inverse:
        w_n = inv(w_n)
    m = 1
    while m < n:
        w_m = pow(w_n, n // (2 * m), P)
        for k in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                t = w * a[k + j + m] % P
                u = a[k + j]
                a[k + j] = (u + t) % P
                a[k + j + m] = (u - t) % P
                w = w * w_m % P
        m *= 2
    if inverse:
        ni = inv(n)
        a = [x * ni % P for x in a]
    return a


def naive_dft(a):
    n = len(a)
    w = root_of_unity(n.bit_length() - 1)
    return [sum(a[j] * pow(w, j * k, P) for j in range(n)) % P for k in range(n)]


def coset_lde(coeffs, rate_bits, shift=GENERATOR):
    """evaluations of the polynomial on shift*<w_N>, N = len(coeffs) << rate_bits, natural order"""
    n = len(coeffs)
    big = [c * pow(shift, i, P) % P for i, c in enumerate(coeffs)] + [0] * ((n << rate_bits) - n)
    return ntt(big)


def eval_poly(coeffs, x):
    r = 0
    for c in reversed(coeffs):
        r = (r * x + c) % P
    return r


# ---- quadratic extension (a0 + a1 X), X^2 = 7
def ext_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def ext_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def ext_mul(a, b):
    return ((a[0] * b[0] + W * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def ext_inv(a):
    d = inv((a[0] * a[0] - W * a[1] * a[1]) % P)
    return (a[0] * d % P, (-a[1]) * d % P)
