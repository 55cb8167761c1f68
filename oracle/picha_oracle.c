/*
 * picha_oracle.c — TEST INFRASTRUCTURE ONLY (see picha_oracle.h).
 *
 * Scalar C restatement of the reference's resize / colour-convert arithmetic.  Every function
 * cites the reference lines (jhs67/picha v0.5.7) whose behaviour it restates.  The evaluation
 * order, the float/int conversions and the rounding points follow the reference exactly; build
 * with -ffp-contract=off so no multiply-add is fused (node-gyp's Release build targets baseline
 * x86-64 and therefore has no FMA either).
 *
 * Parity pinning: tests/test_oracle_golden.py checks this file against the three golden cases the
 * reference's own tests hold for the path (test/color_convert.js:22-40, test/resize.js:20-31).
 * The reference itself cannot be compiled here (src/picha.h:5-7 needs <nan.h>, which the image
 * lacks), so every other format/filter combination is pinned only by this restatement.
 */
#include "picha_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- pixel model: src/picha.h:79-200 -------------------------------------------------------- */

static const int k_bytes[8] = {3, 4, 1, 2, 2, 4, 6, 8};    /* PixelTraits<P>::bytes    picha.h:118-172 */
static const int k_channels[8] = {3, 4, 1, 2, 1, 2, 3, 4}; /* PixelTraits<P>::channels */

int oracle_pixel_bytes(int p) { return (p >= 0 && p < 8) ? k_bytes[p] : 0; }
int oracle_pixel_channels(int p) { return (p >= 0 && p < 8) ? k_channels[p] : 0; }
static int is16(int p) { return p >= 4; }

/* std::min(a, b) = (b < a) ? b : a ; std::max(a, b) = (a < b) ? b : a  (argument order matters for
 * -0.0 / NaN; kept as the reference writes them) */
static inline float std_minf(float a, float b) { return (b < a) ? b : a; }
static inline float std_maxf(float a, float b) { return (a < b) ? b : a; }

/* i::linear_unpack<N,T>  picha.h:98-105:  (*s - i) * (1 / (a - i))  with i = 0.0f, a = float(max) */
static inline __attribute__((always_inline)) void unpack_px(const int pixel, const void *src, float *dst) {
    int n = k_channels[pixel];
    if (is16(pixel)) {
        const uint16_t *s = (const uint16_t *)src;
        const float i = 0.0f, a = 65535.0f;
        for (int k = 0; k < n; ++k) dst[k] = ((float)(int)s[k] - i) * (1 / (a - i));
    } else {
        const uint8_t *s = (const uint8_t *)src;
        const float i = 0.0f, a = 255.0f;
        for (int k = 0; k < n; ++k) dst[k] = ((float)(int)s[k] - i) * (1 / (a - i));
    }
}

/* i::linear_pack<N,T>  picha.h:107-114:  T(max(i, min(a, i + v * (a - i) + 0.5f))) */
static inline __attribute__((always_inline)) void pack_px(const int pixel, const float *src, void *dst) {
    int n = k_channels[pixel];
    if (is16(pixel)) {
        uint16_t *d = (uint16_t *)dst;
        const float i = 0.0f, a = 65535.0f;
        for (int k = 0; k < n; ++k) d[k] = (uint16_t)std_maxf(i, std_minf(a, i + src[k] * (a - i) + 0.5f));
    } else {
        uint8_t *d = (uint8_t *)dst;
        const float i = 0.0f, a = 255.0f;
        for (int k = 0; k < n; ++k) d[k] = (uint8_t)std_maxf(i, std_minf(a, i + src[k] * (a - i) + 0.5f));
    }
}

static inline char *img_row(const oracle_image *im, int y) { /* NativeImage::row picha.h:210 (int arithmetic) */
    return (char *)im->data + y * im->stride;
}

/* ---- filters: src/resize.cc:200-268 ---------------------------------------------------------- */

static float f_triangle(float o) { return 1.0f - fabsf(o); } /* resize.cc:200-203 */
static float f_box(float o) { (void)o; return 1.0f; }        /* resize.cc:205-208 */

/* MitchelFamilyFilter resize.cc:210-231; coefficient expressions evaluated in float, left to right */
static float mitchel_family(float B, float C, float o) {
    float x = fabsf(o);
    if (x < 1) {
        const float A_3 = (12 - 9 * B - 6 * C) / 6;
        const float A_2 = (-18 + 12 * B + 6 * C) / 6;
        const float A_0 = (6 - 2 * B) / 6;
        return A_0 + (x * x * (A_2 + x * A_3));
    } else {
        const float B_3 = (-B - 6 * C) / 6;
        const float B_2 = (6 * B + 30 * C) / 6;
        const float B_1 = (-12 * B - 48 * C) / 6;
        const float B_0 = (8 * B + 24 * C) / 6;
        return B_0 + (x * (B_1 + x * (B_2 + x * B_3)));
    }
}
static float f_catmulrom(float o) { return mitchel_family(0.0f, 0.5f, o); }   /* resize.cc:233-236 */
static float f_mitchel(float o) { return mitchel_family(0.333f, 0.333f, o); } /* resize.cc:238-241 */

/* LanczosFamilyFilter<2> resize.cc:246-255: A * sin(x) * sin(x / A) / x2, A unsigned 2, std::sin(float) */
static float f_lanczos(float o) {
    float x = o * (float)M_PI, x2 = x * x;
    return x2 == 0 ? 1.0f : 2 * sinf(x) * sinf(x / 2) / x2;
}

/* CubicFilter resize.cc:257-260 */
static float f_cubic(float o) {
    o = fabsf(o);
    return 1.0f - o * o * (0.75f - 0.25f * o);
}

typedef float (*filter_fn)(float);
typedef struct scaled_filter { /* ScaledFilter<F> resize.cc:262-268 */
    filter_fn fn;
    float base_support;
    float scale;
} scaled_filter;

static float sf_support(const scaled_filter *f) { return f->scale * f->base_support; }
static float sf_eval(const scaled_filter *f, float v) { return f->fn(v / f->scale) / f->scale; }

/* resizeImage(opts, src, dst) filter switch resize.cc:270-280; tag order resize.cc:151-160 */
static int make_filter(int tag, float scale, scaled_filter *out) {
    switch (tag) {
    case 0: out->fn = f_cubic; out->base_support = 2.0f; break;
    case 1: out->fn = f_lanczos; out->base_support = 2.0f; break; /* float(A), A = 2 */
    case 2: out->fn = f_catmulrom; out->base_support = 2.0f; break;
    case 3: out->fn = f_mitchel; out->base_support = 2.0f; break;
    case 4: out->fn = f_box; out->base_support = 0.5f; break;
    case 5: out->fn = f_triangle; out->base_support = 1.0f; break;
    default: return -3;
    }
    out->scale = scale;
    return 0;
}

/* ---- makeContribs: src/resize.cc:19-50 ----------------------------------------------------- */

typedef struct contrib_range { int left, right, weights; } contrib_range; /* resize.cc:13-15 */

typedef struct fvec { float *p; size_t n, cap; } fvec;
static int fvec_push(fvec *v, float x) {
    if (v->n == v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 1024;
        float *q = (float *)realloc(v->p, nc * sizeof(float));
        if (!q) return -1;
        v->p = q; v->cap = nc;
    }
    v->p[v->n++] = x;
    return 0;
}

static int make_contribs(contrib_range *ranges, int count, const scaled_filter *filter, float scale,
                         fvec *storage, int size) {
    float support = sf_support(filter);
    float fscale = fmaxf(fmaxf(scale, 1.0f), 1.0f / support); /* std::max on finite positives */
    float fsupport = support * fscale;
    float iscale = 1.0f / fscale;

    float center = (float)0.5 * scale;
    for (int i = 0; i < count; ++i, center += scale) {
        float totalweight = 0;
        int left = (int)std_maxf((float)0, ceilf(center - fsupport));
        int right = (int)std_minf((float)(size - 1), floorf(center + fsupport));
        while (left < right && sf_eval(filter, (center - left) * iscale) == 0) left += 1;
        while (right > left && sf_eval(filter, (center - right) * iscale) == 0) right -= 1;
        ranges[i].left = left;
        ranges[i].right = right;
        ranges[i].weights = (int)storage->n;
        for (int j = left; j <= right; ++j) {
            float o = center - (float)j;
            float w = sf_eval(filter, o * iscale);
            if (fvec_push(storage, w)) return -1;
            totalweight += w;
        }
        if (!(totalweight > 0)) return -7; /* assert(totalweight > 0) resize.cc:45 */
        float normalize = (float)1 / totalweight;
        for (int j = ranges[i].weights; j != (int)storage->n; ++j) storage->p[j] *= normalize;
    }
    return 0;
}

int oracle_make_contribs(int filter, float filter_scale, int src_size, int dst_size, int32_t *left,
                         int32_t *right, float *weights, int max_taps) {
    scaled_filter f;
    if (make_filter(filter, filter_scale, &f)) return -3;
    contrib_range *r = (contrib_range *)calloc((size_t)dst_size, sizeof(contrib_range));
    fvec st = {0, 0, 0};
    float scale = src_size / (float)dst_size; /* resize.cc:72-73 */
    int rc = make_contribs(r, dst_size, &f, scale, &st, src_size);
    int maxn = 0;
    if (rc == 0) {
        for (int i = 0; i < dst_size; ++i) {
            int n = r[i].right - r[i].left + 1;
            if (n > maxn) maxn = n;
        }
        if (maxn > max_taps) rc = -10;
    }
    if (rc == 0) {
        memset(weights, 0, sizeof(float) * (size_t)dst_size * (size_t)max_taps);
        for (int i = 0; i < dst_size; ++i) {
            left[i] = r[i].left;
            right[i] = r[i].right;
            for (int j = 0; j <= r[i].right - r[i].left; ++j)
                weights[(size_t)i * max_taps + j] = st.p[r[i].weights + j];
        }
        rc = maxn;
    }
    free(r);
    free(st.p);
    return rc;
}

/* ---- resizeImagePixel: src/resize.cc:66-134 ------------------------------------------------ */

/* The reference instantiates resizeImagePixel<Pixel, Filter> per pixel format (resize.cc:136-149), so its unpack / pack
 * loops have compile-time channel counts and sample types.  This body is force-inlined into one wrapper per format
 * below with `pixel` a literal: same statements, same order, same rounding points — only the dispatch moves out of
 * the per-tap loop (a run-time `pixel` there made this port 2.6x slower than the code it restates). */
static inline __attribute__((always_inline)) int resize_pixel_body(const oracle_image *src, const oracle_image *dst,
                                                                   int filter_tag, float filter_scale, const int pixel) {
    scaled_filter filter;
    if (make_filter(filter_tag, filter_scale, &filter)) return -3;
    const int pixelChannels = k_channels[pixel];
    const int pixelBytes = k_bytes[pixel];

    /* resize.cc:72-79 */
    float xscale = src->width / (float)dst->width;
    float yscale = src->height / (float)dst->height;
    float support = sf_support(&filter);
    float yfscale = fmaxf(fmaxf(yscale, 1.0f), 1.0f / support);
    float yfsupport = support * yfscale;
    int maxycontrib = (int)ceilf(2 * yfsupport);

    /* FloatBuffer tmp(dst.width, maxycontrib, channels) resize.cc:52-64,83: zero-initialised ring */
    int tmp_stride = dst->width * pixelChannels;
    float *tmp = (float *)calloc((size_t)tmp_stride * (size_t)maxycontrib, sizeof(float));
    contrib_range *rowcontribs = (contrib_range *)calloc((size_t)dst->width, sizeof(contrib_range));
    contrib_range *columncontribs = (contrib_range *)calloc((size_t)dst->height, sizeof(contrib_range));
    fvec contribs = {0, 0, 0};
    int rc = 0;
    if (!tmp || !rowcontribs || !columncontribs) { rc = -1; goto done; }

    /* one shared weight store, x table then y table  resize.cc:86-97 */
    rc = make_contribs(rowcontribs, dst->width, &filter, xscale, &contribs, src->width);
    if (rc) goto done;
    rc = make_contribs(columncontribs, dst->height, &filter, yscale, &contribs, src->height);
    if (rc) goto done;

    {
        float centery = (float)0.5 * yscale; /* resize.cc:99-101 */
        int srcrow = (int)std_maxf((float)0, ceilf(centery - yfsupport));
        float unpacked[4];
        for (int y = 0; y < dst->height; ++y, centery += yscale) {
            /* horizontal pass over newly needed source rows  resize.cc:104-119 */
            int needrow = src->height - 1 < (int)(centery + yfsupport) ? src->height - 1 : (int)(centery + yfsupport);
            while (srcrow <= needrow) {
                char *srcdata = img_row(src, srcrow);
                float *tmprow = tmp + (size_t)(srcrow % maxycontrib) * tmp_stride;
                memset(tmprow, 0, sizeof(float) * (size_t)pixelChannels * (size_t)dst->width);
                for (int x = 0; x < dst->width; ++x) {
                    for (int c = rowcontribs[x].left, w = rowcontribs[x].weights; c <= rowcontribs[x].right; ++c, ++w) {
                        unpack_px(pixel, srcdata + c * pixelBytes, unpacked);
                        for (int p = 0; p < pixelChannels; ++p) tmprow[p] += contribs.p[w] * unpacked[p];
                    }
                    tmprow += pixelChannels;
                }
                ++srcrow;
            }
            /* vertical pass + pack  resize.cc:121-132 (ring slot = c % maxycontrib, may alias) */
            char *dstrow = img_row(dst, y);
            for (int x = 0; x < dst->width; ++x) {
                float px[4] = {0, 0, 0, 0};
                for (int c = columncontribs[y].left, w = columncontribs[y].weights; c <= columncontribs[y].right; ++c, ++w) {
                    float *srcpixel = tmp + (size_t)(c % maxycontrib) * tmp_stride + x * pixelChannels;
                    for (int p = 0; p < pixelChannels; ++p) px[p] += contribs.p[w] * srcpixel[p];
                }
                pack_px(pixel, px, dstrow);
                dstrow += pixelBytes;
            }
        }
    }
done:
    free(tmp);
    free(rowcontribs);
    free(columncontribs);
    free(contribs.p);
    return rc;
}

#define ORACLE_RESIZE_PIXEL(P)                                                                                   \
    static __attribute__((noinline)) int resize_pixel_##P(const oracle_image *src, const oracle_image *dst, int filter_tag, \
                                                          float filter_scale) {                                  \
        return resize_pixel_body(src, dst, filter_tag, filter_scale, P);                                         \
    }
ORACLE_RESIZE_PIXEL(0) ORACLE_RESIZE_PIXEL(1) ORACLE_RESIZE_PIXEL(2) ORACLE_RESIZE_PIXEL(3)
ORACLE_RESIZE_PIXEL(4) ORACLE_RESIZE_PIXEL(5) ORACLE_RESIZE_PIXEL(6) ORACLE_RESIZE_PIXEL(7)
#undef ORACLE_RESIZE_PIXEL

/* resizeImage<Filter> 8-way pixel switch  resize.cc:136-149 */
int oracle_resize(const oracle_image *src, const oracle_image *dst, int filter_tag, float filter_scale) {
    if (src->pixel != dst->pixel) return -5; /* assert resize.cc:137 */
    switch (src->pixel) {
    case 0: return resize_pixel_0(src, dst, filter_tag, filter_scale);
    case 1: return resize_pixel_1(src, dst, filter_tag, filter_scale);
    case 2: return resize_pixel_2(src, dst, filter_tag, filter_scale);
    case 3: return resize_pixel_3(src, dst, filter_tag, filter_scale);
    case 4: return resize_pixel_4(src, dst, filter_tag, filter_scale);
    case 5: return resize_pixel_5(src, dst, filter_tag, filter_scale);
    case 6: return resize_pixel_6(src, dst, filter_tag, filter_scale);
    case 7: return resize_pixel_7(src, dst, filter_tag, filter_scale);
    }
    return -1;
}

/* ---- colour convert: src/colorconvert.cc:6-188 ------------------------------------------------ */

void oracle_color_settings(double r, double g, double b, float out[3]) {
    /* ColorSettings() colorconvert.h:11-14 ; getSettings colorconvert.cc:6-22 */
    float rFactor = 0.299f, gFactor = 0.587f, bFactor = (float)0.114;
    if (r == r) rFactor = (float)r;
    if (g == g) gFactor = (float)g;
    if (b == b) bFactor = (float)b;
    float n = 1.0f / (rFactor + gFactor + bFactor);
    rFactor *= n; gFactor *= n; bFactor *= n;
    out[0] = rFactor; out[1] = gFactor; out[2] = bFactor;
}

/* ChannelConvertOp<S,D> colorconvert.cc:24-134 — selected by channel COUNT only */
static inline __attribute__((always_inline)) void channel_convert(const int sc, const int dc, float r, float g, float b, const float *s, float *d) {
    if (sc == dc) { for (int i = 0; i < sc; ++i) d[i] = s[i]; return; }
    switch (sc * 10 + dc) {
    case 12: d[0] = s[0]; d[1] = 1; break;
    case 13: d[0] = s[0]; d[1] = s[0]; d[2] = s[0]; break;
    case 14: d[0] = s[0]; d[1] = s[0]; d[2] = s[0]; d[3] = 1; break;
    case 21: d[0] = s[0]; break;
    case 23: d[0] = s[0]; d[1] = s[1]; d[2] = 0; break;
    case 24: d[0] = s[0]; d[1] = s[0]; d[2] = s[0]; d[3] = s[1]; break;
    case 31: d[0] = s[0] * r + s[1] * g + s[2] * b; break;
    case 32: d[0] = s[0] * r + s[1] * g + s[2] * b; d[1] = 1; break;
    case 34: d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = 1; break;
    case 41: d[0] = s[0] * r + s[1] * g + s[2] * b; break;
    case 42: d[0] = s[0] * r + s[1] * g + s[2] * b; d[1] = s[3]; break;
    case 43: d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; break;
    }
}

/* ColorConverter<Src,Dst>::op colorconvert.cc:136-152, force-inlined per (source, destination) format like the
 * reference's 8 x 8 template dispatch (colorconvert.cc:154-188) */
static inline __attribute__((always_inline)) void convert_body(const oracle_image *src, const oracle_image *dst, float r, float g,
                                                               float b, const int sp, const int dp) {
    const int sb = k_bytes[sp], db = k_bytes[dp];
    const int sc = k_channels[sp], dc = k_channels[dp];
    float srcpack[4], dstpack[4];
    for (int i = 0; i < src->height; ++i) { /* colorconvert.cc:143-150 */
        char *s = img_row(src, i), *d = img_row(dst, i);
        for (int j = 0; j < src->width; ++j, s += sb, d += db) {
            unpack_px(sp, s, srcpack);
            channel_convert(sc, dc, r, g, b, srcpack, dstpack);
            pack_px(dp, dstpack, d);
        }
    }
}
#define ORACLE_CONVERT_SRC(SP)                                                                                         \
    static __attribute__((noinline)) void convert_src_##SP(const oracle_image *src, const oracle_image *dst, float r, \
                                                           float g, float b) {                                        \
        switch (dst->pixel) { /* doSrcColorConvert colorconvert.cc:154-169 */                                          \
        case 0: convert_body(src, dst, r, g, b, SP, 0); break;                                                        \
        case 1: convert_body(src, dst, r, g, b, SP, 1); break;                                                        \
        case 2: convert_body(src, dst, r, g, b, SP, 2); break;                                                        \
        case 3: convert_body(src, dst, r, g, b, SP, 3); break;                                                        \
        case 4: convert_body(src, dst, r, g, b, SP, 4); break;                                                        \
        case 5: convert_body(src, dst, r, g, b, SP, 5); break;                                                        \
        case 6: convert_body(src, dst, r, g, b, SP, 6); break;                                                        \
        case 7: convert_body(src, dst, r, g, b, SP, 7); break;                                                        \
        }                                                                                                             \
    }
ORACLE_CONVERT_SRC(0) ORACLE_CONVERT_SRC(1) ORACLE_CONVERT_SRC(2) ORACLE_CONVERT_SRC(3)
ORACLE_CONVERT_SRC(4) ORACLE_CONVERT_SRC(5) ORACLE_CONVERT_SRC(6) ORACLE_CONVERT_SRC(7)
#undef ORACLE_CONVERT_SRC

int oracle_color_convert(const oracle_image *src, const oracle_image *dst, float r, float g, float b) {
    if (src->width != dst->width || src->height != dst->height) return -5; /* asserts colorconvert.cc:138-139 */
    if (src->pixel < 0 || src->pixel > 7 || dst->pixel < 0 || dst->pixel > 7) return -1;
    if (src->pixel == dst->pixel) { /* doColorConvert colorconvert.cc:172-175 ; NativeImage::copy picha.cc:27-34 */
        int rw = src->width * k_bytes[src->pixel];
        for (int h = 0; h < src->height; ++h) memcpy(img_row(dst, h), img_row(src, h), (size_t)rw);
        return 0;
    }
    switch (src->pixel) { /* doColorConvert colorconvert.cc:176-187 */
    case 0: convert_src_0(src, dst, r, g, b); break;
    case 1: convert_src_1(src, dst, r, g, b); break;
    case 2: convert_src_2(src, dst, r, g, b); break;
    case 3: convert_src_3(src, dst, r, g, b); break;
    case 4: convert_src_4(src, dst, r, g, b); break;
    case 5: convert_src_5(src, dst, r, g, b); break;
    case 6: convert_src_6(src, dst, r, g, b); break;
    case 7: convert_src_7(src, dst, r, g, b); break;
    }
    return 0;
}

/* ---- libuv-pool model for the CPU baseline timing (src/resize.cc:362-364) ------------------ */

typedef struct mt_job {
    const oracle_image *src, *dst;
    int n, filter;
    float scale;
    int next;
    int rc;
    pthread_mutex_t mu;
} mt_job;

static void *mt_worker(void *arg) {
    mt_job *j = (mt_job *)arg;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        int i = j->next++;
        pthread_mutex_unlock(&j->mu);
        if (i >= j->n) break;
        int rc = oracle_resize(&j->src[i], &j->dst[i], j->filter, j->scale);
        if (rc) { pthread_mutex_lock(&j->mu); j->rc = rc; pthread_mutex_unlock(&j->mu); }
    }
    return 0;
}

int oracle_resize_batch_mt(int n, const oracle_image *src, const oracle_image *dst, int filter,
                           float filter_scale, int threads) {
    mt_job j;
    j.src = src; j.dst = dst; j.n = n; j.filter = filter; j.scale = filter_scale; j.next = 0; j.rc = 0;
    pthread_mutex_init(&j.mu, 0);
    if (threads < 1) threads = 1;
    pthread_t *t = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    for (int i = 0; i < threads; ++i) pthread_create(&t[i], 0, mt_worker, &j);
    for (int i = 0; i < threads; ++i) pthread_join(t[i], 0);
    free(t);
    pthread_mutex_destroy(&j.mu);
    return j.rc;
}
