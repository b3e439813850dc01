// PNG decode for the batch drivers (missing 4): `read_image(path, grayscale)` =
// `cv2.imread` (imcui/hloc/utils/io.py:11-21) reads whatever OpenCV reads, and the reference's own evaluation fixtures
// (imcui/datasets/wxbs_benchmark/**.png: 88 RGB + 4 gray files, 8 bits, not interlaced) are PNG.
//
// A PNG is a zlib stream of FILTERED scan lines.  Like the JPEG path the work splits at the entropy coder:
//   * host threads of the library (C++, through the C ABI, no interpreter lock): chunk walk, the IDAT chunks through zlib's `inflate`
//     (serial per file by construction) into caller-provided -- pinned -- staging, the filter-type bytes validated;
//   * device: the scan-line filters undone (None / Sub / Up / Average / Paeth, RFC 2083 section 6) and the pixels converted to what the
//     host reader returns (alpha dropped, palette looked up).  A pixel depends on its left, upper and upper-left neighbours, so an image
//     is a 2-D recurrence: rows are serial, and within a row the Average / Paeth chains are serial as well.  It runs as a WAVEFRONT: one
//     workgroup per image, one thread per row, row r working on pixel s - r in step s, one barrier per step; a row hands its newest pixel
//     to the row below through a double-buffered LDS word.  W + rows steps per strip of <= 1024 rows instead of W x H dependent steps.
// Byte work with integer arithmetic: HBM / latency bound, nothing here wants a matrix core.  Bit-exact by construction (the filters are
// exact modulo-256 arithmetic); the checker is PIL (tests/test_png_cpu.py, tests/test_gpu_png.py).
// Not taken (the caller keeps its host reader): interlaced files, bit depths other than 8, files with an eXIf orientation other than 1.
// Chunk CRCs are not verified (zlib's adler32 over the pixel stream is).
#include <string.h>
#include <zlib.h>

#include <atomic>
#include <thread>
#include <vector>

#include "common.h"
#include "imcui_hip.h"

#define PI_W 0
#define PI_H 1
#define PI_CT 2     // PNG colour type: 0 gray, 2 RGB, 3 palette, 4 gray + alpha, 6 RGBA
#define PI_CIN 3    // samples per pixel in the file
#define PI_COUT 4   // channels of the decoded image: 1 (gray sources) or 3
#define PI_BPP 5    // bytes per pixel in the file (= PI_CIN at 8 bits)
#define PI_NPAL 6   // palette entries
#define PI_INTS 8

namespace {

struct Png {
    int W = 0, H = 0, ct = 0, cin = 0, cout = 0, bpp = 0, npal = 0;
    unsigned char pal[768];
    std::vector<std::pair<const unsigned char*, size_t>> idat;
};

// Orientation (tag 0x0112) of an eXIf chunk body = a TIFF header + IFD0; 1 when absent or unreadable.  cv2.imread and PIL's
// `exif_transpose` rotate / mirror by it; the device path does not, so anything but 1 is handed back to the host reader.
int exif_orientation(const unsigned char* b, size_t n) {
    if (n < 8) return 1;
    const bool le = b[0] == 'I' && b[1] == 'I';
    if (!le && !(b[0] == 'M' && b[1] == 'M')) return 1;
    auto u16 = [&](size_t o) -> unsigned { return le ? (b[o] | (b[o + 1] << 8)) : ((b[o] << 8) | b[o + 1]); };
    auto u32 = [&](size_t o) -> unsigned { return le ? (u16(o) | (u16(o + 2) << 16)) : ((u16(o) << 16) | u16(o + 2)); };
    if (u16(2) != 42) return 1;
    const size_t ifd = u32(4);
    if (ifd + 2 > n) return 1;
    const unsigned cnt = u16(ifd);
    for (unsigned e = 0; e < cnt; ++e) {
        const size_t o = ifd + 2 + 12 * (size_t)e;
        if (o + 12 > n) return 1;
        if (u16(o) == 0x0112) return u16(o + 2) == 3 ? (int)u16(o + 8) : 1;
    }
    return 1;
}

inline unsigned be32(const unsigned char* p) { return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | p[3]; }

int png_parse(const unsigned char* d, size_t n, Png& p, bool want_idat) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (n < 8 + 25 || memcmp(d, sig, 8) != 0) return IMCUI_ERR_ARG;
    memset(p.pal, 0, sizeof p.pal);
    size_t i = 8;
    bool have_ihdr = false, have_idat = false;
    while (i + 12 <= n) {
        const size_t len = be32(d + i);
        const unsigned char* type = d + i + 4;
        if (len > n || i + 12 + len > n) return IMCUI_ERR_ARG;
        const unsigned char* body = d + i + 8;
        if (!have_ihdr) {
            if (memcmp(type, "IHDR", 4) != 0 || len != 13) return IMCUI_ERR_ARG;
            const unsigned W = be32(body), H = be32(body + 4);
            const int depth = body[8], ct = body[9], comp = body[10], filt = body[11], lace = body[12];
            if (W == 0 || H == 0 || W > 65535 || H > 65535 || comp != 0 || filt != 0) return IMCUI_ERR_ARG;
            if (ct != 0 && ct != 2 && ct != 3 && ct != 4 && ct != 6) return IMCUI_ERR_ARG;
            if (lace != 0 || depth != 8) return IMCUI_ERR_UNSUPPORTED;  // Adam7 / 1-2-4-16 bit samples: the host reader
            p.W = (int)W;
            p.H = (int)H;
            p.ct = ct;
            p.cin = ct == 0 ? 1 : ct == 2 ? 3 : ct == 3 ? 1 : ct == 4 ? 2 : 4;
            p.bpp = p.cin;
            p.cout = (ct == 0 || ct == 4) ? 1 : 3;
            have_ihdr = true;
        } else if (memcmp(type, "PLTE", 4) == 0) {
            if (len % 3 != 0 || len > 768) return IMCUI_ERR_ARG;
            memcpy(p.pal, body, len);
            p.npal = (int)(len / 3);
        } else if (memcmp(type, "eXIf", 4) == 0) {
            const int o = exif_orientation(body, len);
            if (o != 1 && o >= 0 && o <= 8) return IMCUI_ERR_UNSUPPORTED;  // rotated / mirrored on load by cv2 and PIL: the host reader
        } else if (memcmp(type, "IDAT", 4) == 0) {
            have_idat = true;
            if (want_idat) p.idat.emplace_back(body, len);
        } else if (memcmp(type, "IEND", 4) == 0) {
            break;
        }
        i += 12 + len;
    }
    if (!have_ihdr || !have_idat) return IMCUI_ERR_ARG;
    if (p.ct == 3 && p.npal == 0) return IMCUI_ERR_ARG;
    return IMCUI_OK;
}

void png_fill_info(const Png& p, int* info) {
    memset(info, 0, PI_INTS * sizeof(int));
    info[PI_W] = p.W;
    info[PI_H] = p.H;
    info[PI_CT] = p.ct;
    info[PI_CIN] = p.cin;
    info[PI_COUT] = p.cout;
    info[PI_BPP] = p.bpp;
    info[PI_NPAL] = p.npal;
}

int png_inflate_one(const unsigned char* d, size_t n, unsigned char* raw, size_t raw_bytes, unsigned char* pal) {
    Png p;
    const int rc = png_parse(d, n, p, true);
    if (rc != IMCUI_OK) return rc;
    const size_t stride = (size_t)p.W * p.bpp + 1, need = stride * p.H;
    if (need != raw_bytes) return IMCUI_ERR_ARG;
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit(&z) != Z_OK) return IMCUI_ERR_ARG;
    size_t done = 0;
    int zr = Z_OK;
    for (size_t k = 0; k < p.idat.size() && zr != Z_STREAM_END; ++k) {
        z.next_in = const_cast<unsigned char*>(p.idat[k].first);
        z.avail_in = (uInt)p.idat[k].second;
        while (z.avail_in > 0 && zr != Z_STREAM_END) {
            // Once the image is full a valid stream still has its trailer (adler32) to consume, possibly across an IDAT boundary: it gets a
            // scratch to write into and is refused only if it actually produces a byte there, or stops making progress.
            unsigned char spill[32];
            const size_t room = need - done;
            z.next_out = room > 0 ? raw + done : spill;
            z.avail_out = room > 0 ? (uInt)(room > 0x40000000u ? 0x40000000u : room) : (uInt)sizeof spill;
            const uInt before_out = z.avail_out, before_in = z.avail_in;
            zr = inflate(&z, Z_NO_FLUSH);
            const uInt produced = before_out - z.avail_out;
            if (room > 0) done += produced;
            if (zr != Z_OK && zr != Z_STREAM_END) break;
            if ((room == 0 && produced > 0) || (produced == 0 && z.avail_in == before_in && zr == Z_OK)) {
                zr = Z_DATA_ERROR;  // more data than the image holds / no progress
                break;
            }
        }
        if (zr != Z_OK && zr != Z_STREAM_END) break;
    }
    inflateEnd(&z);
    if (zr != Z_STREAM_END || done != need) return IMCUI_ERR_ARG;  // truncated / corrupt stream (zlib checks its adler32 at the end)
    for (int r = 0; r < p.H; ++r)
        if (raw[(size_t)r * stride] > 4) return IMCUI_ERR_ARG;  // filter type
    if (pal) memcpy(pal, p.pal, 768);
    return IMCUI_OK;
}

}  // namespace

// info [PI_INTS ints]: width, height, colour type, samples per pixel, channels of the decoded image (1 or 3), bytes per pixel, palette entries
extern "C" int imcui_hip_png_info(const unsigned char* data, size_t n, int* info) {
    if (!data || !info) return IMCUI_ERR_ARG;
    Png p;
    const int rc = png_parse(data, n, p, false);
    if (rc != IMCUI_OK) return rc;
    png_fill_info(p, info);
    return IMCUI_OK;
}
// bytes of the filtered scan lines of a file (what imcui_hip_png_inflate writes): H x (1 + W x bytes per pixel)
extern "C" size_t imcui_hip_png_raw_bytes(const int* info) { return info ? (size_t)info[PI_H] * ((size_t)info[PI_W] * info[PI_BPP] + 1) : 0; }

// raw [host, raw_bytes]: the filtered scan lines (filter-type byte + W x bpp bytes per row); palette [host, 768] or NULL.  Re-entrant.
extern "C" int imcui_hip_png_inflate(const unsigned char* data, size_t n, unsigned char* raw, size_t raw_bytes, unsigned char* palette) {
    if (!data || !raw) return IMCUI_ERR_ARG;
    return png_inflate_one(data, n, raw, raw_bytes, palette);
}
// `count` files on `threads` host threads of the library; raw[i] / raw_bytes[i]: destination of file i (typically slices of one pinned staging
// buffer), palettes [count][768], status [count] per-file return code (a refused file does not stop the others)
extern "C" int imcui_hip_png_inflate_batch(const unsigned char* const* data, const size_t* sizes, int count, unsigned char* const* raw, const size_t* raw_bytes,
                                           unsigned char* palettes, int* status, int threads) {
    if (!data || !sizes || !raw || !raw_bytes || !status || count < 0) return IMCUI_ERR_ARG;
    if (threads < 1) threads = 1;
    if (threads > count) threads = count > 0 ? count : 1;
    std::atomic<int> next(0);
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= count) return;
            status[i] = (data[i] && raw[i]) ? png_inflate_one(data[i], sizes[i], raw[i], raw_bytes[i], palettes ? palettes + 768 * (size_t)i : nullptr) : IMCUI_ERR_ARG;
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    return IMCUI_OK;
}

// ------------------------------------------------------------------------------------------------ device side
struct PngJob {
    const unsigned char* raw;  // filtered scan lines
    unsigned char* out;        // [H][W][cout]
    const unsigned char* pal;  // [256][3] (colour type 3)
    unsigned* lastrow;         // [W] the last row of a strip, for images taller than a workgroup
    int W, H, ct, bpp;
};
#define PNG_JOBS 24  // jobs per launch (kernel arguments: no device-side table, no copy)
struct PngJobs {
    PngJob j[PNG_JOBS];
};

__device__ __forceinline__ int png_paeth(int a, int b, int c) {
    const int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

__global__ __launch_bounds__(1024) void png_unfilter_kernel(PngJobs jobs) {
    __shared__ unsigned exch[2][1024];
    const PngJob j = jobs.j[blockIdx.x];
    const int tid = threadIdx.x, NT = blockDim.x;
    const size_t stride = (size_t)j.W * j.bpp + 1;
    const int cout = (j.ct == 0 || j.ct == 4) ? 1 : 3;
    for (int r0 = 0; r0 < j.H; r0 += NT) {
        const int r = r0 + tid;
        const bool act = r < j.H;
        const int rows = min(NT, j.H - r0);
        const unsigned char* line = j.raw + (size_t)(act ? r : 0) * stride;
        const int ft = act ? line[0] : 0;
        const bool more = r0 + NT < j.H;  // another strip follows: its first row needs this strip's last row
        unsigned left = 0, upleft = 0;
        const int nsteps = j.W + rows - 1;
        for (int s = 0; s < nsteps; ++s) {
            const int i = s - tid;
            if (act && i >= 0 && i < j.W) {
                const unsigned char* fp = line + 1 + (size_t)i * j.bpp;
                unsigned f = 0;
                for (int c = 0; c < j.bpp; ++c) f |= (unsigned)fp[c] << (8 * c);
                unsigned up = 0;
                if (r > 0) up = (tid == 0) ? j.lastrow[i] : exch[(s - 1) & 1][tid - 1];
                unsigned x = 0;
                for (int c = 0; c < j.bpp; ++c) {
                    const int fb = (f >> (8 * c)) & 255, a = (left >> (8 * c)) & 255, b = (up >> (8 * c)) & 255, cc = (upleft >> (8 * c)) & 255;
                    const int pred = ft == 1 ? a : ft == 2 ? b : ft == 3 ? ((a + b) >> 1) : ft == 4 ? png_paeth(a, b, cc) : 0;
                    x |= (unsigned)((fb + pred) & 255) << (8 * c);
                }
                exch[s & 1][tid] = x;
                if (more && tid == rows - 1) j.lastrow[i] = x;
                unsigned char* o = j.out + ((size_t)r * j.W + i) * cout;
                if (j.ct == 3) {
                    const unsigned char* q = j.pal + 3 * (x & 255);
                    o[0] = q[0], o[1] = q[1], o[2] = q[2];
                } else if (cout == 1) {
                    o[0] = (unsigned char)(x & 255);  // gray, gray + alpha (alpha dropped)
                } else {
                    o[0] = (unsigned char)(x & 255), o[1] = (unsigned char)((x >> 8) & 255), o[2] = (unsigned char)((x >> 16) & 255);  // RGB, RGBA (alpha dropped)
                }
                upleft = up;
                left = x;
            }
            __syncthreads();
        }
    }
}

// workspace: one row of packed pixels per image taller than 1024 rows
extern "C" size_t imcui_hip_png_workspace_bytes(const int* infos, int count) {
    size_t t = 256;
    for (int i = 0; infos && i < count; ++i) t += align_up((size_t)infos[i * PI_INTS + PI_W] * 4, 256);
    return t;
}

// raw [dev] + raw_offsets [host, count]: the filtered scan lines of file i at raw + raw_offsets[i]; infos [host, count x PI_INTS];
// palettes [dev, count x 768] (may be NULL when no file has a palette); out [dev] + out_offsets [host]: [H][W][channels] uint8 of file i.
extern "C" int imcui_hip_png_reconstruct_batch(imcui_hip_t* h, const unsigned char* raw, const size_t* raw_offsets, const int* infos, const unsigned char* palettes,
                                               int count, unsigned char* out, const size_t* out_offsets, void* ws, size_t ws_bytes, void* stream_) {
    if (!h || !raw || !raw_offsets || !infos || !out || !out_offsets) return imcui_set_err(h, IMCUI_ERR_ARG, "png: null argument");
    if (count <= 0) return IMCUI_OK;
    if (!ws || ws_bytes < imcui_hip_png_workspace_bytes(infos, count)) return imcui_set_err(h, IMCUI_ERR_WS, "png: workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    WsAlloc a(ws, ws_bytes);
    for (int i0 = 0; i0 < count; i0 += PNG_JOBS) {
        PngJobs jobs;
        memset(&jobs, 0, sizeof jobs);
        const int m = count - i0 < PNG_JOBS ? count - i0 : PNG_JOBS;
        int hmax = 0;
        for (int k = 0; k < m; ++k) {
            const int* inf = infos + (size_t)(i0 + k) * PI_INTS;
            PngJob& j = jobs.j[k];
            j.W = inf[PI_W], j.H = inf[PI_H], j.ct = inf[PI_CT], j.bpp = inf[PI_BPP];
            if (j.W <= 0 || j.H <= 0 || j.bpp < 1 || j.bpp > 4 || (j.ct == 3 && !palettes)) return imcui_set_err(h, IMCUI_ERR_ARG, "png: bad info record of file %d", i0 + k);
            j.raw = raw + raw_offsets[i0 + k];
            j.out = out + out_offsets[i0 + k];
            j.pal = palettes ? palettes + 768 * (size_t)(i0 + k) : nullptr;
            j.lastrow = a.get<unsigned>((size_t)j.W);
            hmax = j.H > hmax ? j.H : hmax;
        }
        const int nt = hmax >= 1024 ? 1024 : ((hmax + 63) / 64) * 64;
        hipLaunchKernelGGL(png_unfilter_kernel, dim3(m), dim3(nt), 0, stream, jobs);
    }
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}
