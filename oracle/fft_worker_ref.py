"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's worker-level transform operators and of the block loop that
calls them.  Nothing under pil2-stark-js_amd/ may import this file.

  interpolatePrepareBlock   src/helpers/fft/fft_worker.js:6-19
  _fft_block / fft_block    src/helpers/fft/fft_worker.js:21-67
  BR, traspose, bitReverse, invBitReverse, _fft   src/helpers/fft/fft_p.js:10-64, 114-176

Pure-Python loops over Python ints (small cases only).  fft_worker.js needs `workerpool`, which is absent here, so no vector can
be generated from it directly; the restatement is pinned through the transform it composes: fft_p(...) below, built from
fft_block exactly as fft_p.js:114-176 does, must reproduce the outputs of the reference's scalar fft / ifft (fft/fft.js, loaded
by oracle/gen_golden.js -> tests/golden/ntt.json) for every block size (tests/test_fft_worker.py).
"""
P = 0xFFFFFFFF00000001
SHIFT = 7


def root(k):
    """F.w[k] (f3g.js:40 buildFFT(this, 7277203076849721926n), fft/fft.js:39-50): w[32] is that constant, w[k] = w[k+1]^2"""
    w = 7277203076849721926
    for _ in range(32 - k):
        w = w * w % P
    return w


def interpolatePrepareBlock(buff, width, start, inc):
    """fft_worker.js:6-19, in place; buff: list of ints, height x width row-major"""
    height = len(buff) // width
    w = start % P
    for i in range(height):
        for j in range(width):
            buff[i * width + j] = buff[i * width + j] * w % P
        w = w * inc % P
    return buff


def _fft_block(buff, rel_pos, start_pos, nPols, nBits, s, blockBits, layers):
    """fft_worker.js:21-60"""
    n, m = 1 << nBits, 1 << blockBits
    md2 = m >> 1
    if layers < blockBits:                                                    # :30-34
        _fft_block(buff, rel_pos, start_pos, nPols, nBits, s, blockBits - 1, layers)
        _fft_block(buff, rel_pos, start_pos + md2, nPols, nBits, s, blockBits - 1, layers)
        return
    if layers > 1:                                                            # :35-38
        _fft_block(buff, rel_pos, start_pos, nPols, nBits, s - 1, blockBits - 1, layers - 1)
        _fft_block(buff, rel_pos, start_pos + md2, nPols, nBits, s - 1, blockBits - 1, layers - 1)
    if s > blockBits:                                                         # :40-50
        width = 1 << (s - layers)
        heigth = n // width
        y, x = start_pos // heigth, start_pos % heigth
        w = pow(root(s), x * width + y, P)
    else:
        w = 1
    wl = root(layers)
    for i in range(md2):                                                      # :52-59
        for j in range(nPols):
            a, b = (start_pos - rel_pos + i) * nPols + j, (start_pos - rel_pos + md2 + i) * nPols + j
            t = w * buff[b] % P
            u = buff[a] % P
            buff[a] = (u + t) % P
            buff[b] = (u - t) % P
        w = w * wl % P


def fft_block(buff, start_pos, nPols, nBits, s, blockBits, layers):
    """fft_worker.js:62-67"""
    _fft_block(buff, start_pos, start_pos, nPols, nBits, s, blockBits, layers)
    return buff


def BR(x, nBits):
    """fft_p.js:10-17"""
    return int(format(x, "0%db" % nBits)[::-1], 2) if nBits else 0


def traspose(dst, src, nPols, nBits, trasposeBits):
    """fft_p.js:20-32"""
    n, w = 1 << nBits, 1 << trasposeBits
    h = n // w
    for i in range(w):
        for j in range(h):
            fi, di = j * w + i, i * h + j
            dst[di * nPols:(di + 1) * nPols] = src[fi * nPols:(fi + 1) * nPols]


def fft_p(src, nPols, nBits, inverse, blockBits, block_op=fft_block):
    """_fft (fft_p.js:114-176) with the block size handed in (the reference derives it from the worker count and clamps it to
    [12, 16] and to nBits, :125-129) and the block operator replaceable (the device twin in the GPU tests).  -> new list"""
    n = 1 << nBits
    blockBits = min(nBits, blockBits)
    blockSize = 1 << blockBits
    nBlocks = n // blockSize
    a = [0] * (n * nPols)
    if inverse:                                                               # invBitReverse, fft_p.js:54-64
        nInv = pow(n, P - 2, P)
        for i in range(n):
            rii = (n - BR(i, nBits)) % n
            for p_ in range(nPols):
                a[i * nPols + p_] = src[rii * nPols + p_] * nInv % P
    else:                                                                     # bitReverse, fft_p.js:35-42
        for i in range(n):
            ri = BR(i, nBits)
            a[i * nPols:(i + 1) * nPols] = src[ri * nPols:(ri + 1) * nPols]
    b = [0] * (n * nPols)
    i = 0
    while i < nBits:                                                          # :153-173
        sInc = min(blockBits, nBits - i)
        for j in range(nBlocks):
            bb = a[j * blockSize * nPols:(j + 1) * blockSize * nPols]
            bb = block_op(bb, j * blockSize, nPols, nBits, i + sInc, blockBits, sInc)
            a[j * blockSize * nPols:(j + 1) * blockSize * nPols] = bb
        if sInc < nBits:
            traspose(b, a, nPols, nBits, sInc)
            a, b = b, a
        i += blockBits
    return a


def _rounds(a, nPols, nBits, blockBits, block_op):
    """the block rounds shared by _fft and interpolate (fft_p.js:153-173, 238-260, 268-292); a is consumed"""
    n = 1 << nBits
    blockBits = min(nBits, blockBits)
    blockSize = 1 << blockBits
    b = [0] * (n * nPols)
    i = 0
    while i < nBits:
        sInc = min(blockBits, nBits - i)
        for j in range(n // blockSize):
            bb = a[j * blockSize * nPols:(j + 1) * blockSize * nPols]
            a[j * blockSize * nPols:(j + 1) * blockSize * nPols] = block_op(bb, j * blockSize, nPols, nBits, i + sInc, blockBits, sInc)
        if sInc < nBits:
            traspose(b, a, nPols, nBits, sInc)
            a, b = b, a
        i += blockBits
    return a


def interpolate_p(src, nPols, nBits, nBitsExt, blockBits, blockBitsExt, nPerThread, block_op=fft_block, prepare_op=interpolatePrepareBlock):
    """interpolate (fft_p.js:187-297) with the block sizes and the rows per prepare call handed in (the reference derives them from
    the worker count, :199-203, :212-216, :66-84).  -> new list of 2^nBitsExt x nPols"""
    n, extN = 1 << nBits, 1 << nBitsExt
    a = [0] * (n * nPols)
    for i in range(n):                                                        # interpolateBitReverse, fft_p.js:44-52
        rii = (n - BR(i, nBits)) % n
        a[i * nPols:(i + 1) * nPols] = src[rii * nPols:(rii + 1) * nPols]
    a = _rounds(a, nPols, nBits, blockBits, block_op)
    invN = pow(n, P - 2, P)                                                   # interpolatePrepare, fft_p.js:66-107
    for i in range(0, n, nPerThread):
        curN = min(nPerThread, n - i)
        bb = a[i * nPols:(i + curN) * nPols]
        a[i * nPols:(i + curN) * nPols] = prepare_op(bb, nPols, invN * pow(SHIFT, i, P) % P, SHIFT)
    a = a + [0] * ((extN - n) * nPols)
    e = [0] * (extN * nPols)
    for i in range(extN):                                                     # bitReverse at the extended size, :265
        ri = BR(i, nBitsExt)
        e[i * nPols:(i + 1) * nPols] = a[ri * nPols:(ri + 1) * nPols]
    return _rounds(e, nPols, nBitsExt, blockBitsExt, block_op)
