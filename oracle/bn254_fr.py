"""BN254 scalar field Fr and its NTT.  TEST INFRASTRUCTURE.

The FFT lives in gnark-crypto (ecc/bn254/fr/fft, un-vendored: gnark-plonky2-verifier/go.mod:9; used inside `groth16.Prove`,
gnark-plonky2-verifier/cmd/web-api.go:77): restated from its definition -- values[k] = sum_j coeffs[j] w^(jk) with
w = ROOT_OF_UNITY^(2^28 / n); coset = multiply coefficient j by GENERATOR^j first (fft.OnCoset).  r as in
contracts/hardhat/contracts/Verifier.sol:34.  ROOT_OF_UNITY / GENERATOR are gnark-crypto's constants [UPSTREAM]; their
defining properties (order exactly 2^28; 5 a non-residue) are asserted below.  PARITY UNPINNED against gnark's output order
(the reference holds no FFT vector): checked by definition (naive DFT), inverse and coset-evaluation properties.
Memory format of the C ABI = gnark-crypto's: x * 2^256 mod r as 4 little-endian u64.
"""
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
ROOT_OF_UNITY = 19103219067921713944291392827692070036145651957329286315305642004821462161904
TWO_ADICITY = 28
GENERATOR = 5
assert pow(ROOT_OF_UNITY, 1 << 28, R) == 1 and pow(ROOT_OF_UNITY, 1 << 27, R) == R - 1
assert pow(GENERATOR, (R - 1) // 2, R) == R - 1


def root(log_n):
    return pow(ROOT_OF_UNITY, 1 << (TWO_ADICITY - log_n), R)


def ntt(a, inverse=False, coset=False):
    """natural order in and out"""
    n = len(a)
    log_n = n.bit_length() - 1
    a = [x % R for x in a]
    if coset and not inverse:
        a = [x * pow(GENERATOR, i, R) % R for i, x in enumerate(a)]
    bits = log_n
    a = [a[int(format(i, "0%db" % bits)[::-1], 2) if bits else 0] for i in range(n)]
    w_n = root(log_n)
    if inverse:
        w_n = pow(w_n, R - 2, R)
    m = 1
    while m < n:
        w_m = pow(w_n, n // (2 * m), R)
        for k in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                t = w * a[k + j + m] % R
                u = a[k + j]
                a[k + j], a[k + j + m] = (u + t) % R, (u - t) % R
                w = w * w_m % R
        m *= 2
    if inverse:
        ni = pow(n, R - 2, R)
        a = [x * ni % R for x in a]
        if coset:
            gi = pow(GENERATOR, R - 2, R)
            a = [x * pow(gi, i, R) % R for i, x in enumerate(a)]
    return a


def naive_dft(a, shift=1):
    n = len(a)
    w = root(n.bit_length() - 1)
    return [sum(c * pow(shift * pow(w, k, R) % R, j, R) for j, c in enumerate(a)) % R for k in range(n)]


def to_mont_words(x):
    m = x * (1 << 256) % R
    return [(m >> (64 * i)) & (2**64 - 1) for i in range(4)]


def from_mont_words(w):
    m = sum(int(w[i]) << (64 * i) for i in range(4))
    return m * pow(1 << 256, R - 2, R) % R
