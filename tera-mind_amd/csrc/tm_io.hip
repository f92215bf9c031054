// Tile I/O either side of the denoising path.
//
//  * gene_tile_scatter: the COO transcript tile of utils/MBADataset_tst.py:65-91 (`_getgene`: gblk x gblk
//    block sum + z padding; `_pad_gn`: shift by the halo offset and crop to the gsz x gsz grid) and the
//    densification of model/unet_ours.py:301-306, as one scatter-add pass.  Counts are integers held in
//    fp32, so the atomic adds are exact and order independent (bit-exact against the oracle).
//  * blosc_decompress: decoder of the Blosc-1 frames that zarr 2.14.1 / numcodecs 0.15.0 (the reference's
//    pins, environment.yml:172,220) write by default for the per-step state tiles (test_brn.py:225
//    `zarr.save_array`): lz4 codec, byte shuffle.  Host code; restated from the published c-blosc 1.x
//    frame layout (16-byte header, bstarts, per-split streams) and the LZ4 block format.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/teramind_hip.h"
#include "tm_kernels.h"

namespace tmk {

// one thread per COO entry; grid-stride so that nnz up to 2^31 is covered with a bounded grid
__global__ __launch_bounds__(256) void gene_tile_scatter_kernel(const int32_t* __restrict__ crd, const float* __restrict__ dat,
                                                                long nnz, int gblk, int shift_h, int shift_w, int gsz,
                                                                int chan_in, int zpad_ch, float* __restrict__ out) {
  const long ch_out = (long)chan_in + 2L * zpad_ch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (long)gridDim.x * blockDim.x) {
    const int h = crd[i], w = crd[nnz + i], c = crd[2 * nnz + i];
    if (h < 0 || w < 0 || c < 0 || c >= chan_in) continue;      // outside the declared shape: ignored, never written
    const int gh = h / gblk + shift_h, gw = w / gblk + shift_w;
    if (gh < 0 || gh >= gsz || gw < 0 || gw >= gsz) continue;   // _pad_gn crop
    atomicAdd(out + ((long)gh * gsz + gw) * ch_out + zpad_ch + c, dat[i]);
  }
}

hipError_t launch_gene_tile_scatter(const int32_t* crd, const float* dat, long nnz, int gblk, int shift_h, int shift_w,
                                    int gsz, int chan_in, int zpad_ch, float* out, hipStream_t s) {
  const size_t bytes = (size_t)gsz * gsz * ((size_t)chan_in + 2 * (size_t)zpad_ch) * sizeof(float);
  hipError_t e = hipMemsetAsync(out, 0, bytes, s);
  if (e != hipSuccess || nnz == 0) return e;
  long blocks = (nnz + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  gene_tile_scatter_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(crd, dat, nnz, gblk, shift_h, shift_w, gsz, chan_in,
                                                                       zpad_ch, out);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// LZ4 block format: sequences of [token][literal length ext][literals][offset LE16][match length ext].
// Returns the number of bytes written, or -1 on malformed input / overflow of `cap`.
static long lz4_block_decode(const uint8_t* src, long n, uint8_t* dst, long cap) {
  long ip = 0, op = 0;
  while (ip < n) {
    const unsigned tok = src[ip++];
    long lit = tok >> 4;
    if (lit == 15) {
      unsigned b;
      do {
        if (ip >= n) return -1;
        b = src[ip++];
        lit += b;
      } while (b == 255);
    }
    if (ip + lit > n || op + lit > cap) return -1;
    memcpy(dst + op, src + ip, (size_t)lit);
    ip += lit;
    op += lit;
    if (ip >= n) break;                       // the last sequence carries literals only
    if (ip + 2 > n) return -1;
    const long off = src[ip] | (src[ip + 1] << 8);
    ip += 2;
    if (off == 0 || off > op) return -1;
    long ml = (tok & 15);
    if (ml == 15) {
      unsigned b;
      do {
        if (ip >= n) return -1;
        b = src[ip++];
        ml += b;
      } while (b == 255);
    }
    ml += 4;
    if (op + ml > cap) return -1;
    for (long k = 0; k < ml; ++k) dst[op + k] = dst[op + k - off];      // overlapping copies are the RLE case
    op += ml;
  }
  return op;
}

static inline uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

enum { BLOSC_SHUFFLE = 1, BLOSC_MEMCPYED = 2, BLOSC_BITSHUFFLE = 4, BLOSC_DONT_SPLIT = 16 };
static const int BLOSC_MAX_SPLITS = 16, BLOSC_MIN_BUFFERSIZE = 128, BLOSC_HEADER = 16;

int blosc_decompress(const void* src_, size_t src_bytes, void* dst_, size_t dst_cap, size_t* out_bytes) {
  const uint8_t* src = (const uint8_t*)src_;
  uint8_t* dst = (uint8_t*)dst_;
  if (!src || src_bytes < (size_t)BLOSC_HEADER) return io_fail(TM_ERR_ARG, "blosc: frame shorter than its header");
  const unsigned flags = src[2];
  const long typesize = src[3] ? src[3] : 1;
  const long nbytes = le32(src + 4), blocksize = le32(src + 8), cbytes = le32(src + 12);
  if (out_bytes) *out_bytes = (size_t)nbytes;
  if (!dst) return TM_OK;                                        // size query
  if ((size_t)cbytes > src_bytes) return io_fail(TM_ERR_ARG, "blosc: frame truncated");
  if ((size_t)nbytes > dst_cap) return io_fail(TM_ERR_ARG, "blosc: destination too small");
  if (nbytes == 0) return TM_OK;
  if (flags & BLOSC_MEMCPYED) {
    if (cbytes < BLOSC_HEADER + nbytes) return io_fail(TM_ERR_ARG, "blosc: memcpyed frame truncated");
    memcpy(dst, src + BLOSC_HEADER, (size_t)nbytes);
    return TM_OK;
  }
  if (flags & BLOSC_BITSHUFFLE) return io_fail(TM_ERR_ARG, "blosc: bit-shuffled frames are not supported (zarr default is byte shuffle)");
  const int codec = (flags >> 5) & 7;
  if (codec != 1) return io_fail(TM_ERR_ARG, "blosc: only the lz4 codec is supported (zarr/numcodecs default)");
  if (blocksize <= 0) return io_fail(TM_ERR_ARG, "blosc: bad block size");
  const long nblocks = (nbytes + blocksize - 1) / blocksize, leftover = nbytes % blocksize;
  if (BLOSC_HEADER + 4 * nblocks > cbytes) return io_fail(TM_ERR_ARG, "blosc: block table truncated");
  std::vector<uint8_t> tmp((size_t)blocksize);
  const bool shuffled = (flags & BLOSC_SHUFFLE) && typesize > 1;
  for (long j = 0; j < nblocks; ++j) {
    const bool last_short = (j == nblocks - 1) && leftover > 0;
    const long bsize = last_short ? leftover : blocksize;
    long nsplits = 1;
    if (!(flags & BLOSC_DONT_SPLIT) && typesize <= BLOSC_MAX_SPLITS && bsize / typesize >= BLOSC_MIN_BUFFERSIZE && !last_short)
      nsplits = typesize;
    const long neblock = bsize / nsplits;
    long ip = le32(src + BLOSC_HEADER + 4 * j);
    uint8_t* o = shuffled ? tmp.data() : dst + j * blocksize;
    long done = 0;
    for (long sp = 0; sp < nsplits; ++sp) {
      if (ip + 4 > cbytes) return io_fail(TM_ERR_ARG, "blosc: split header out of range");
      const long cb = le32(src + ip);
      ip += 4;
      if (cb < 0 || ip + cb > cbytes) return io_fail(TM_ERR_ARG, "blosc: split out of range");
      if (cb == neblock) {
        memcpy(o + done, src + ip, (size_t)neblock);
      } else if (lz4_block_decode(src + ip, cb, o + done, neblock) != neblock) {
        return io_fail(TM_ERR_ARG, "blosc: lz4 stream does not decode to the split size");
      }
      ip += cb;
      done += neblock;
    }
    if (done != bsize) return io_fail(TM_ERR_ARG, "blosc: block size mismatch");
    if (shuffled) {
      uint8_t* d = dst + j * blocksize;
      const long nel = bsize / typesize, rem = bsize - nel * typesize;
      for (long k = 0; k < typesize; ++k) {
        const uint8_t* s = tmp.data() + k * nel;
        for (long i = 0; i < nel; ++i) d[i * typesize + k] = s[i];
      }
      memcpy(d + nel * typesize, tmp.data() + nel * typesize, (size_t)rem);
    }
  }
  return TM_OK;
}

}  // namespace tmk
