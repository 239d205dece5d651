// model.hip -- plan builder (Keras build order of the reference's layers), step orchestration and the C ABI.
//
// Reference structure followed (annotator/models/tf_models): components.py:16-81 Downsample, :84-166 Upsample,
// :169-247 Encoder, :250-320 Decoder; unet.py:19-88 UNet, :91-191 MulmoUNet, :194-300 annotators.
#include "model.h"

#include <rccl/rccl.h>
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <cmath>

#include "fast.h"
#include "kernels.h"

namespace dnnca {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

enum { kScalars = 8 };

Model::~Model() {
    fast_release(this);
    ig_release(this);
    for (void* a : allocs) (void)hipFree(a);
    for (auto& r : recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto e : evpool) (void)hipEventDestroy(e);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (comm) ncclCommDestroy(comm);
    if (ev_bucket) (void)hipEventDestroy(ev_bucket);
    if (ev_comm_done) (void)hipEventDestroy(ev_comm_done);
    if (comm_stream) (void)hipStreamDestroy(comm_stream);
    for (auto& sl : stage) {
        if (sl.uploaded) (void)hipEventDestroy(sl.uploaded);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (out_ring) (void)hipHostFree(out_ring);
    if (aug_pin) (void)hipHostFree(aug_pin);
    for (auto e : aug_ev)
        if (e) (void)hipEventDestroy(e);
    if (wg_fork) (void)hipEventDestroy(wg_fork);
    if (wg_join) (void)hipEventDestroy(wg_join);
    if (wg_bucket) (void)hipEventDestroy(wg_bucket);
    if (wg_stream) (void)hipStreamDestroy(wg_stream);
    if (stream) (void)hipStreamDestroy(stream);
}

bool Model::wg_side_begin() {
    const bool off = getenv("DNNCA_NO_WG_STREAM") != nullptr;          // read per call: the tests flip it
    // full profiles (modes 1, 3) time one launch after the other (the sampled bracket of mode 2 goes where its kernel goes); the
    // dry run launches nothing.  (With a communicator a gradient bucket also waits for the side stream: send_bucket.)
    if (off || dry || prof_mode == 1 || prof_mode == 3) return false;
    if (!wg_stream) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);           // lo: the numerically largest = least urgent
        const char* pe = getenv("DNNCA_WG_PRIO");                   // tuning aid: 0 normal, 1 least urgent (default), 2 most urgent
        const int prio = !pe || atoi(pe) == 1 ? lo : (atoi(pe) == 2 ? hi : 0);
        if (hipStreamCreateWithPriority(&wg_stream, hipStreamNonBlocking, prio) != hipSuccess) { wg_stream = nullptr; return false; }
        if (hipEventCreateWithFlags(&wg_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&wg_join, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&wg_bucket, hipEventDisableTiming) != hipSuccess)
            return false;
    }
    if (!wg_fork || !wg_join || !wg_bucket) return false;
    if (hipEventRecord(wg_fork, stream) != hipSuccess || hipStreamWaitEvent(wg_stream, wg_fork, 0) != hipSuccess) return false;
    wg_pending = true;
    stream = wg_stream;
    return true;
}

void Model::wg_side_end(hipStream_t main) { stream = main; }

int Model::wg_side_join() {
    if (!wg_pending) return DNNCA_OK;
    wg_pending = false;
    HIP_TRY(hipEventRecord(wg_join, wg_stream));
    HIP_TRY(hipStreamWaitEvent(stream, wg_join, 0));
    return DNNCA_OK;
}

int Model::alloc(void** ptr, size_t bytes) {
    if (bytes == 0) bytes = 4;
    HIP_TRY(hipMalloc(ptr, bytes));
    allocs.push_back(*ptr);
    HIP_TRY(hipMemsetAsync(*ptr, 0, bytes, stream));
    return DNNCA_OK;
}

void Model::set_variant(const char* fmt, ...) {
    char buf[64];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    variant = buf;
}

// ---------------------------------------------------------------------------------------------- launch accounting
bool Model::begin(const char* name, double bytes, double flops) {
    if (dry) {
        char line[256];
        snprintf(line, sizeof(line), "%s%s%s\t%.0f\t%.0f\n", name, variant.empty() ? "" : "#", variant.c_str(), bytes, flops);
        plan_text += line;
        variant.clear();
        return false;
    }
    if (prof_mode == 0) return true;
    if (prof_mode == 2 && (focus != name || (prof_period > 1 && iterations % prof_period != 0))) return true;
    int id;
    std::string key = name;
    if (prof_mode == 3 && cur_op) key += "@" + *cur_op;          // per-layer table
    auto it = kid.find(key);
    if (it == kid.end()) {
        id = (int)kstats.size();
        kid[key] = id;
        KStat ks;
        ks.name = key;
        kstats.push_back(ks);
    } else {
        id = it->second;
    }
    KStat& ks = kstats[id];
    ks.bytes += bytes;
    ks.flops += flops;
    ks.launches += 1;
    Rec r;
    r.id = id;
    auto get = [&]() {
        hipEvent_t e = nullptr;
        if (!evpool.empty()) {
            e = evpool.back();
            evpool.pop_back();
        } else {
            (void)hipEventCreate(&e);
        }
        return e;
    };
    r.a = get();
    r.b = get();
    (void)hipEventRecord(r.a, stream);
    recs.push_back(r);
    return true;
}

void Model::end() {
    if (dry || prof_mode == 0 || recs.empty()) return;
    Rec& r = recs.back();
    if (r.b && r.id >= 0) {
        (void)hipEventRecord(r.b, stream);
        r.id = -r.id - 1;   // mark closed
    }
}

int Model::flush_profile() {
    if (recs.empty()) return DNNCA_OK;
    HIP_TRY(hipStreamSynchronize(stream));
    for (auto& r : recs) {
        int id = r.id < 0 ? -r.id - 1 : r.id;
        float ms = 0.f;
        if (r.id < 0 && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) kstats[id].total_ms += ms;
        evpool.push_back(r.a);
        evpool.push_back(r.b);
    }
    recs.clear();
    return DNNCA_OK;
}

// ---------------------------------------------------------------------------------------------- plan
int Model::build() {
    const dnnca_model_desc& d = desc;
    if (d.arch != DNNCA_ARCH_UNET && d.arch != DNNCA_ARCH_MULMO) { set_error("unknown arch %d", d.arch); return DNNCA_EINVAL; }
    if (d.conv_stride != 1) { set_error("conv_stride %d unsupported (reference configs use 1)", d.conv_stride); return DNNCA_EINVAL; }
    if (d.padding != DNNCA_PAD_SAME) { set_error("padding: valid is unsupported (every reference config uses same; labels could not match the cropped logits in utils/losses.py:34)"); return DNNCA_EINVAL; }
    if (d.kernel_size < 1 || d.kernel_size % 2 == 0) { set_error("kernel_size %d must be odd", d.kernel_size); return DNNCA_EINVAL; }
    if (d.in_channels < 1 || d.n_filters_first < 1 || d.n_downsample < 1 || d.rate < 2 || d.n_conv < 1 || d.max_batch < 1) { set_error("bad model_options"); return DNNCA_EINVAL; }
    if (d.dtype != DNNCA_F32 && d.dtype != DNNCA_BF16) { set_error("unknown dtype"); return DNNCA_EINVAL; }
    int div = 1;
    for (int i = 0; i < d.n_downsample; ++i) div *= d.rate;
    if (d.height % div || d.width % div) { set_error("H, W (%d, %d) must be divisible by rate^n_downsample = %d", d.height, d.width, div); return DNNCA_EINVAL; }
    const bool mulmo = d.arch == DNNCA_ARCH_MULMO;
    const int nenc = mulmo ? d.in_channels : 1;
    if (mulmo && (d.reference_index < 0 || d.reference_index >= nenc)) { set_error("reference_index out of range"); return DNNCA_EINVAL; }
    const int K = d.kernel_size, r = d.rate, L = d.n_downsample;
    const size_t MB = (size_t)d.max_batch;
    const float act = d.leaky_alpha;   // 0 relu, >0 leaky

    auto add_param = [&](const std::string& name, std::initializer_list<int64_t> shape, int trainable) -> int64_t {
        ParamInfo pi;
        pi.name = name;
        pi.ndim = (int)shape.size();
        pi.size = 1;
        int i = 0;
        for (int k = 0; k < 4; ++k) pi.shape[k] = 1;
        for (auto s : shape) {
            pi.shape[i++] = s;
            pi.size *= s;
        }
        pi.trainable = trainable;
        int64_t& n = trainable ? nT : nS;
        pi.offset = n;
        n += pi.size;
        params.push_back(pi);
        return pi.offset;
    };

    std::vector<std::pair<float**, size_t>> pending;   // tensors to allocate: (slot, floats)
    struct Slot { float* d; float* g; };
    std::vector<Slot*> slots;
    // tensors are allocated immediately (stream-ordered memset) -- simple and the sizes are static
    int rc = DNNCA_OK;
    auto new_tensor = [&](int H, int W, int C) -> T {
        T t;
        size_t n = MB * H * W * C;
        float *dp = nullptr, *gp = nullptr;
        if (rc == DNNCA_OK) rc = alloc((void**)&dp, n * 4);
        if (rc == DNNCA_OK) rc = alloc((void**)&gp, n * 4);
        t.d.p = dp; t.d.H = H; t.d.W = W; t.d.C = C; t.d.ps = C;
        t.g = t.d;
        t.g.p = gp;
        return t;
    };

    auto add_bn = [&](const std::string& prefix, const T& in, const T& out) {
        Op o;
        o.type = OP_BN;
        o.name = prefix;
        o.inA = in;
        o.out = out;
        int C = in.d.C;
        o.w_off = add_param(prefix + ".gamma", {C}, 1);
        o.b_off = add_param(prefix + ".beta", {C}, 1);
        o.mm_off = add_param(prefix + ".moving_mean", {C}, 0);
        o.mv_off = add_param(prefix + ".moving_variance", {C}, 0);
        if (rc == DNNCA_OK) rc = alloc((void**)&o.coef, (size_t)4 * C * 4);
        if (rc == DNNCA_OK) rc = alloc((void**)&o.ws, (size_t)2 * C * 8);
        ops.push_back(o);
    };
    auto add_conv = [&](const std::string& prefix, const T& inA, const T& inB, const T& out, bool need_din) {
        Op o;
        o.type = OP_CONV;
        o.name = prefix;
        o.inA = inA;
        o.inB = inB;
        o.out = out;
        o.k = K;
        o.alpha = act;
        o.need_din = need_din;
        o.w_off = add_param(prefix + ".kernel", {K, K, inA.d.C + inB.d.C, out.d.C}, 1);
        o.b_off = add_param(prefix + ".bias", {out.d.C}, 1);
        ops.push_back(o);
    };

    xin.d.H = d.height; xin.d.W = d.width; xin.d.C = d.in_channels; xin.d.ps = d.in_channels;
    xin.g = xin.d;
    std::vector<int> filters;
    for (int i = 0, f = d.n_filters_first; i < L; ++i, f *= r) filters.push_back(f);

    T empty;   // C = 0
    std::vector<std::vector<T>> skips(nenc);
    std::vector<T> bottoms(nenc);
    int hb = d.height / div, wb = d.width / div;
    T bottom_cat;
    if (mulmo) bottom_cat = new_tensor(hb, wb, filters[L - 1] * nenc);
    for (int e = 0; e < nenc; ++e) {
        std::string enc = mulmo ? "encoder" + std::to_string(e) : "encoder";
        T cur = mulmo ? tslice(xin, e, 1) : xin;
        bool first = true;
        int H = d.height, W = d.width;
        for (int i = 0; i < L; ++i) {
            std::string p = enc + ".down" + std::to_string(i);
            int f = filters[i];
            for (int j = 0; j < d.n_conv; ++j) {
                T a = new_tensor(H, W, f);
                add_conv(p + ".conv" + std::to_string(j), cur, empty, a, !first);
                first = false;
                cur = a;
                if (d.bn) {
                    T n = new_tensor(H, W, f);
                    add_bn(p + ".bn" + std::to_string(j), cur, n);
                    cur = n;
                }
            }
            skips[e].push_back(cur);
            H /= r;
            W /= r;
            bool last = (i == L - 1);
            T dest = (mulmo && last) ? tslice(bottom_cat, e * f, f) : T();
            T pooled = (d.bn || !(mulmo && last)) ? new_tensor(H, W, f) : dest;
            Op o;
            o.type = OP_POOL;
            o.name = p + ".pool";
            o.inA = cur;
            o.out = pooled;
            o.k = r;
            ops.push_back(o);
            cur = pooled;
            if (d.bn) {
                T n = (mulmo && last) ? dest : new_tensor(H, W, f);
                add_bn(p + ".pool_bn", cur, n);
                cur = n;
            }
        }
        bottoms[e] = cur;
    }
    T cur = mulmo ? bottom_cat : bottoms[0];
    const std::vector<T>& ref = skips[mulmo ? d.reference_index : 0];
    for (int u = 0; u < L; ++u) {
        int i = L - 1 - u;
        int f = filters[i];
        std::string p = "decoder.up" + std::to_string(u);
        int H = cur.d.H * r, W = cur.d.W * r;
        T t = new_tensor(H, W, f);
        Op o;
        o.type = OP_TCONV;
        o.name = p + ".tconv";
        o.inA = cur;
        o.out = t;
        o.k = r;
        o.w_off = add_param(p + ".tconv.kernel", {r, r, f, cur.d.C}, 1);
        o.b_off = add_param(p + ".tconv.bias", {f}, 1);
        ops.push_back(o);
        cur = t;
        if (d.bn) {
            T n = new_tensor(H, W, f);
            add_bn(p + ".tconv_bn", cur, n);
            cur = n;
        }
        T skip = ref[i];
        for (int j = 0; j < d.n_conv; ++j) {
            T a = new_tensor(H, W, f);
            add_conv(p + ".conv" + std::to_string(j), cur, j == 0 ? skip : empty, a, true);
            cur = a;
            if (d.bn) {
                T n = new_tensor(H, W, f);
                add_bn(p + ".bn" + std::to_string(j), cur, n);
                cur = n;
            }
        }
    }
    {
        Op o;
        o.type = OP_HEAD;
        o.name = "head";
        o.inA = cur;
        o.w_off = add_param("head.kernel", {1, 1, cur.d.C, 1}, 1);
        o.b_off = add_param("head.bias", {1}, 1);
        ops.push_back(o);
        outH = cur.d.H;
        outW = cur.d.W;
    }
    if (rc != DNNCA_OK) return rc;

    // backward accumulation flags: the first op (in backward order) to produce a gradient overwrites, later ones add.
    std::map<float*, bool> written;
    for (int i = (int)ops.size() - 1; i >= 0; --i) {
        Op& o = ops[i];
        if (!o.need_din) continue;
        o.accA = written[o.inA.g.p];
        written[o.inA.g.p] = true;
        if (o.type == OP_CONV && o.inB.d.C) {
            o.accB = written[o.inB.g.p];
            written[o.inB.g.p] = true;
        }
    }

    // who reads a BatchNorm's output, and which BatchNorm feeds a conv input (exact tensors only: channel slices do not count)
    for (int b = 0; b < (int)ops.size(); ++b) {
        if (ops[b].type != OP_BN) continue;
        const View& y = ops[b].out.d;
        for (int i = 0; i < (int)ops.size(); ++i) {
            Op& o = ops[i];
            const bool a = o.inA.d.p == y.p && o.inA.d.C == y.C && o.inA.d.ps == y.ps;
            const bool bb = o.type == OP_CONV && o.inB.d.C && o.inB.d.p == y.p && o.inB.d.C == y.C && o.inB.d.ps == y.ps;
            const bool overlap = !a && !bb && ((o.inA.d.p >= y.p && o.inA.d.p < y.p + y.ps) || (o.type == OP_CONV && o.inB.d.C && o.inB.d.p >= y.p && o.inB.d.p < y.p + y.ps));
            if (a || bb || overlap) ops[b].out_readers.push_back(overlap ? -1 : i);          // -1: a reader this analysis does not understand
            if (o.type == OP_CONV && a) o.src_bn[0] = b;
            if (bb) o.src_bn[1] = b;
        }
    }
    // twin structure of the mulmo encoders (model.h: lockstep)
    enc_ops = n_enc = 0;
    if (mulmo && nenc > 1) {
        size_t first_dec = 0;
        while (first_dec < ops.size() && ops[first_dec].name.compare(0, 7, "encoder") == 0) ++first_dec;
        if (first_dec % nenc == 0 && first_dec > 0) {
            const int per = (int)(first_dec / nenc);
            bool same = true;
            for (int e = 1; e < nenc && same; ++e)
                for (int j = 0; j < per && same; ++j) {
                    const Op &a = ops[j], &b = ops[(size_t)e * per + j];
                    same = a.type == b.type && a.k == b.k && a.inA.d.C == b.inA.d.C && a.inA.d.H == b.inA.d.H && a.inA.d.W == b.inA.d.W &&
                           a.inB.d.C == b.inB.d.C && a.out.d.C == b.out.d.C && a.out.d.H == b.out.d.H && a.out.d.W == b.out.d.W;
                }
            if (same) { enc_ops = per; n_enc = nenc; }
        }
    }
    fast_plan_masks(this);
    DN_TRY(ig_plan_half(this));

    // flat buffers
    DN_TRY(alloc((void**)&p, (size_t)nT * 4));
    DN_TRY(alloc((void**)&g, (size_t)(nT + 8 + 4) * 4));
    DN_TRY(alloc((void**)&m, (size_t)nT * 4));
    DN_TRY(alloc((void**)&v, (size_t)nT * 4));
    DN_TRY(alloc((void**)&state, (size_t)nS * 4));
    DN_TRY(alloc((void**)&scalars, kScalars * 8));
    out5 = g + nT;
    size_t npix = MB * outH * outW;
    DN_TRY(alloc((void**)&x_stage, MB * d.height * d.width * d.in_channels * 4));
    DN_TRY(alloc((void**)&y_stage, npix * 4));
    DN_TRY(alloc((void**)&logits, npix * 4));
    DN_TRY(alloc((void**)&dlogits, npix * 4));
    DN_TRY(alloc((void**)&prob, npix * 4));
    DN_TRY(alloc((void**)&thr_dev, DNNCA_CONF_MAX_THR * 4));
    DN_TRY(alloc((void**)&head_partials, 2048 * 72 * 4));
    DN_TRY(alloc((void**)&conf_dev, 2 * (DNNCA_CONF_MAX_THR + 1) * 8));     // confusion histogram / host all-reduce staging
    // Keras defaults for the non-trainable / BN variables: gamma 1, moving_variance 1 (the rest 0)
    {
        std::vector<float> hp((size_t)nT, 0.f), hs((size_t)nS, 0.f);
        for (auto& pi : params) {
            bool one = pi.name.size() > 6 && (pi.name.rfind(".gamma") == pi.name.size() - 6 ||
                                              pi.name.rfind(".moving_variance") == pi.name.size() - 16);
            if (!one) continue;
            float* dst = pi.trainable ? hp.data() : hs.data();
            for (int64_t k = 0; k < pi.size; ++k) dst[pi.offset + k] = 1.f;
        }
        if (nT) HIP_TRY(hipMemcpyAsync(p, hp.data(), (size_t)nT * 4, hipMemcpyHostToDevice, stream));
        if (nS) HIP_TRY(hipMemcpyAsync(state, hs.data(), (size_t)nS * 4, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    return DNNCA_OK;
}

// ---------------------------------------------------------------------------------------------- forward
static inline double nelem(int B, const View& v) { return (double)B * v.H * v.W * v.C; }

// the untuned kernels read and write f32 only: a tensor stored as bf16 (View::h) must have been taken by a tuned kernel
static bool all_f32(const Op& o) {
    if (!(o.inA.d.h | o.inA.g.h | o.inB.d.h | o.inB.g.h | o.out.d.h | o.out.g.h)) return true;
    set_error("internal: %s has a bf16-stored operand but no tuned kernel took it", o.name.c_str());
    return false;
}

bool Model::lockstep() const {
    // OFF by default: nothing shares a launch across the encoders yet, and the order alone costs cache locality (pass_order).
    // DNNCA_LOCKSTEP=1 walks the encoders in lockstep (the groundwork for one launch per twin-op triple; results are unchanged).
    static const bool on = getenv("DNNCA_LOCKSTEP") != nullptr;
    return on && enc_ops > 0 && !(desc.flags & 1);
}

std::vector<int> Model::pass_order(bool backward) const {
    std::vector<int> ord;
    const int n = (int)ops.size(), ne = lockstep() && !(backward && bucketing) ? enc_ops * n_enc : 0;
    // Only the levels whose tensors are small walk in lockstep: at full resolution an op's output (134 MB at 8 x 512 x 512 x 16) is
    // what the next op of the SAME encoder reads, largely out of the 256 MB Infinity Cache; with the other encoders' ops in between it
    // comes from HBM.  Measured on mulmo_unet (8.93 ms sequential, same box): lockstep over all four levels 9.23 ms, levels <= 256^2
    // 9.14, <= 128^2 9.00, <= 64^2 9.02 -- so only the two deep levels are candidates for shared launches.
    static const int max_h = getenv("DNNCA_LOCKSTEP_MAXH") ? atoi(getenv("DNNCA_LOCKSTEP_MAXH")) : 128;
    int j0 = 0;          // first op (relative index) of the lockstep part
    while (ne && j0 < enc_ops && ops[j0].out.d.H > max_h) ++j0;
    if (!backward) {
        for (int e = 0; e < (ne ? n_enc : 0); ++e)
            for (int j = 0; j < j0; ++j) ord.push_back(e * enc_ops + j);
        for (int j = j0; j < (ne ? enc_ops : 0); ++j)
            for (int e = 0; e < n_enc; ++e) ord.push_back(e * enc_ops + j);
        for (int i = ne; i < n; ++i) ord.push_back(i);
    } else {
        for (int i = n - 1; i >= ne; --i) ord.push_back(i);
        for (int j = (ne ? enc_ops : 0) - 1; j >= j0; --j)
            for (int e = n_enc - 1; e >= 0; --e) ord.push_back(e * enc_ops + j);
        for (int e = (ne ? n_enc : 0) - 1; e >= 0; --e)
            for (int j = j0 - 1; j >= 0; --j) ord.push_back(e * enc_ops + j);
    }
    return ord;
}

int Model::forward(const float* x_dev, int B, bool training) {
    if (B < 1 || B > desc.max_batch) { set_error("batch %d outside [1, max_batch=%d]", B, desc.max_batch); return DNNCA_EINVAL; }
    last_batch = B;
    const bool generic = desc.flags & 1;
    // input ops carry a channel offset relative to xin; rebase them on this batch
    for (Op& o : ops)
        if (o.type == OP_CONV && !o.need_din) {
            ptrdiff_t off = o.inA.d.p - xin.d.p;
            o.inA.d.p = const_cast<float*>(x_dev) + off;
        }
    xin.d.p = const_cast<float*>(x_dev);
    step_init_done = false;
    DN_TRY(fast_prepare(this));
    DN_TRY(ig_prepare(this));
    head_in_conv.done = false;
    tail_done = first_done = tconv_done = nullptr;
    fold_deferred = false;
    // the head will ride in the last conv's epilogue: it needs the label statistics (positive rate) of this step, so they go first
    // (pg_prep has just zeroed the scalars)
    bool head_in_conv_ok = head_in_conv.requested && !generic && (step_init_done || dry) && head_defer_ok && ops.size() >= 2 &&
                           ops.back().type == OP_HEAD && fast_head_supported(this, ops.back());
    // ... either from the first encoder block's fused launch (when the head-in-conv kernel will be there to consume its partials
    // table; the label image is then the network's output geometry: the first block runs at full resolution) or from their own kernel
    label_part_valid = false;
    const float* labels_in_first_block = nullptr;
    if (head_in_conv_ok) {
        static const bool fuse_labels = getenv("DNNCA_NO_LABEL_FUSION") == nullptr;
        if (fuse_labels && fast_head_in_conv_possible(this) && ops[0].type == OP_CONV && !ops[0].need_din && ops[0].out.d.H == outH &&
            ops[0].out.d.W == outW)
            labels_in_first_block = head_in_conv.y;
        else
            head_in_conv_ok = fast_label_stats(this, (size_t)B * outH * outW, head_in_conv.y);
    }
    head_in_conv.labels_done = head_in_conv_ok;

    // op_done[i]: op i's work rode in an earlier launch of this pass (a max-pool in the preceding conv's / BatchNorm's launch, the
    // later layers of a fused block, the twin ops of the other mulmo encoders)
    const std::vector<int> order = pass_order(false);
    op_done.assign(ops.size(), 0);
    for (size_t ok_ = 0; ok_ < order.size(); ++ok_) {
        size_t oi = (size_t)order[ok_];
        if (op_done[oi]) continue;
        Op& o = ops[oi];
        cur_op = &o.name;
        switch (o.type) {
            case OP_CONV: {
                int Cin = o.inA.d.C + o.inB.d.C;
                double bytes = 4.0 * (nelem(B, o.inA.d) + nelem(B, o.inB.d) + nelem(B, o.out.d));
                double flops = 2.0 * B * o.out.d.H * o.out.d.W * o.k * o.k * Cin * o.out.d.C;
                if (head_in_conv_ok && oi + 2 == ops.size()) {                // the conv that feeds the head: head + loss + head backward in its epilogue
                    const float gs = (float)(1.0 / ((double)outH * outW * B));
                    const double npx = (double)B * outH * outW;
                    if (fast_tail3(this, B, o, ops.back(), head_in_conv.y, head_in_conv.cfg, gs) ||
                        fast_conv_fwd_head(this, B, o, ops.back(), head_in_conv.y, head_in_conv.cfg, gs, bytes + 4.0 * npx * (1 + o.out.d.C), flops + 30.0 * npx)) {
                        head_in_conv.done = true;
                        break;
                    }
                    if (label_part_valid) { set_error("internal: label partials without the head-in-conv kernel"); return DNNCA_ESTATE; }
                }
                if (oi == 0 && labels_in_first_block) {
                    if (!generic && fused_down_fwd(this, B, oi, training, labels_in_first_block)) {
                        op_done[oi + 1] = op_done[oi + 2] = 1;
                        break;
                    }
                    // the first block did not take them: the label statistics get their own launch after all
                    head_in_conv_ok = fast_label_stats(this, (size_t)B * outH * outW, head_in_conv.y);
                    head_in_conv.labels_done = head_in_conv_ok;
                    labels_in_first_block = nullptr;
                }
                if (!generic && fused_down_fwd(this, B, oi, training)) {      // conv, conv, pool of one encoder block in one launch
                    op_done[oi + 1] = op_done[oi + 2] = 1;
                    break;
                }
                if (!generic && o.inB.d.C && fused_up2_fwd(this, B, oi, training)) {      // the two convs of a decoder block (its transposed conv rode earlier)
                    op_done[oi + 1] = 1;
                    break;
                }
                Op* pool = (!generic && oi + 1 < ops.size() && fast_pool_fusable(this, o, ops[oi + 1])) ? &ops[oi + 1] : nullptr;
                if (!generic && fast_conv_fwd(this, B, o, bytes + (pool ? 4.0 * nelem(B, pool->out.d) : 0.0), flops, pool)) {
                    if (pool) op_done[pool - ops.data()] = 1;
                    break;
                }
                Op* bn_next = (training && oi + 1 < ops.size() && ops[oi + 1].type == OP_BN && ops[oi + 1].inA.d.p == o.out.d.p &&
                               fast_bn_supported(this, ops[oi + 1])) ? &ops[oi + 1] : nullptr;
                if (!generic && (fast_first_conv_fwd(this, B, o, bytes, flops, bn_next) || ig_conv_fwd(this, B, o, bytes, flops, bn_next))) break;
                if (!all_f32(o)) return DNNCA_ESTATE;
                LAUNCH(this, "g_conv_fwd", bytes, flops,
                       g_conv_fwd(stream, B, o.inA.d, o.inB.d, p + o.w_off, p + o.b_off, o.out.d, o.k, o.alpha));
                break;
            }
            case OP_BN: {
                int C = o.inA.d.C;
                double n = (double)B * o.inA.d.H * o.inA.d.W;
                double tb = 4.0 * nelem(B, o.inA.d);
                Op* bn_pool = (!generic && oi + 1 < ops.size() && fast_bn_pool_fusable(this, o, ops[oi + 1])) ? &ops[oi + 1] : nullptr;
                // every reader a conv that normalises while it stages its operands: no apply pass, no normalised tensor
                o.elided = false;
                if (!generic && !bn_pool && !o.out_readers.empty() && fast_bn_supported(this, o)) {
                    bool all = true;
                    for (int r : o.out_readers) all = all && r >= 0 && ops[r].type == OP_CONV && ig_norm_on_load_ok(this, B, ops[r]);
                    o.elided = all;
                }
                Op* pool_bn = (bn_pool && training && oi + 2 < ops.size() && ops[oi + 2].type == OP_BN &&
                               ops[oi + 2].inA.d.p == bn_pool->out.d.p && fast_bn_supported(this, ops[oi + 2])) ? &ops[oi + 2] : nullptr;
                if (!generic && fast_bn_fwd(this, B, o, training, kBnMomentum, kBnEps, bn_pool, pool_bn)) {
                    if (bn_pool) op_done[bn_pool - ops.data()] = 1;
                    break;
                }
                if (!all_f32(o)) return DNNCA_ESTATE;
                if (training) {
                    if (!dry) HIP_TRY(hipMemsetAsync(o.ws, 0, (size_t)2 * C * 8, stream));
                    LAUNCH(this, "g_bn_stats_mean", tb, tb / 4, g_bn_stats_mean(stream, B, o.inA.d, o.ws));
                    LAUNCH(this, "g_bn_stats_var", tb, tb / 2, g_bn_stats_var(stream, B, o.inA.d, o.ws));
                }
                LAUNCH(this, "g_bn_finalize", 0, 0,
                       g_bn_finalize(stream, C, n, o.ws, p + o.w_off, p + o.b_off, state + o.mm_off, state + o.mv_off,
                                     o.coef, training ? 1 : 0, kBnMomentum, kBnEps));
                LAUNCH(this, "g_bn_apply", 2 * tb, tb / 2, g_bn_apply(stream, B, o.inA.d, o.out.d, o.coef));
                break;
            }
            case OP_POOL: {
                // (a pool computed by the launch that produced its input never gets here: op_done)
                double bytes = 4.0 * (nelem(B, o.inA.d) + nelem(B, o.out.d));
                if (!generic && fast_pool_fwd(this, B, o, bytes)) break;
                if (!all_f32(o)) return DNNCA_ESTATE;
                LAUNCH(this, "g_pool_fwd", bytes, 0, g_pool_fwd(stream, B, o.inA.d, o.out.d, o.k));
                break;
            }
            case OP_TCONV: {
                double bytes = 4.0 * (nelem(B, o.inA.d) + nelem(B, o.out.d));
                double flops = 2.0 * nelem(B, o.out.d) * o.inA.d.C;
                int used = 3;
                if (!generic && fused_up_fwd(this, B, oi, training, &used)) { // tconv, conv, conv of one decoder block in one launch
                    for (int t = 1; t < used; ++t) op_done[oi + t] = 1;       // (+ the next block's tconv when it rides along)
                    break;
                }
                if (!generic && fast_up3_fwd(this, B, oi)) {                  // tconv + two-source conv of the 3-channel level
                    op_done[oi + 1] = 1;
                    break;
                }
                Op* bn_next = (training && oi + 1 < ops.size() && ops[oi + 1].type == OP_BN && ops[oi + 1].inA.d.p == o.out.d.p &&
                               fast_bn_supported(this, ops[oi + 1])) ? &ops[oi + 1] : nullptr;
                if (!generic && (fast_tconv_fwd(this, B, o, bytes, flops) || ig_tconv_fwd(this, B, o, bytes, flops, bn_next))) break;
                if (!all_f32(o)) return DNNCA_ESTATE;
                LAUNCH(this, "g_tconv_fwd", bytes, flops,
                       g_tconv_fwd(stream, B, o.inA.d, p + o.w_off, p + o.b_off, o.out.d, o.k));
                break;
            }
            case OP_HEAD: {
                double npix = (double)B * outH * outW;
                head_deferred = false;
                if (defer_head && !generic && fast_head_supported(this, o)) {
                    head_deferred = true;   // runs fused with the loss and its own backward (fast_head_train)
                    break;
                }
                if (!all_f32(o)) return DNNCA_ESTATE;
                LAUNCH(this, "g_head_fwd", 4.0 * (nelem(B, o.inA.d) + npix), 2.0 * nelem(B, o.inA.d),
                       g_head_fwd(stream, B, o.inA.d, p + o.w_off, p + o.b_off, logits));
                break;
            }
        }
    }
    return DNNCA_OK;
}

// ---------------------------------------------------------------------------------------------- loss + backward
int Model::loss_and_backward(const float* y_dev, int B, const dnnca_loss_cfg& cfg, bool backward) {
    cur_op = nullptr;
    if (!(backward && head_in_conv.done && y_dev == head_in_conv.y))
        head_pending.partials = nullptr;    // leftovers of a step that stopped on an error (not: the head that just ran in the forward pass)
    fin_pending.on = false;
    if (cfg.label_smoothing) {          // utils/losses.py:62-67: every later use of y_true (positive rate, assertions, loss) sees the blurred labels
        if (!y_smooth) DN_TRY(alloc((void**)&y_smooth, (size_t)desc.max_batch * outH * outW * 4));
        if (!fast_label_smooth(this, B, outH, outW, y_dev, y_smooth, cfg.label_smoothing_filter_size, cfg.label_smoothing_sigma)) {
            set_error("label_smoothing: filter size %d / sigma %g not supported for %dx%d labels", cfg.label_smoothing_filter_size,
                      (double)cfg.label_smoothing_sigma, outH, outW);
            return DNNCA_EINVAL;
        }
        y_dev = y_smooth;
    }
    const bool generic = desc.flags & 1;
    size_t npix = (size_t)B * outH * outW;
    const bool labels_done = backward && head_in_conv.labels_done && y_dev == head_in_conv.y;
    const bool head_was_in_conv = labels_done && head_in_conv.done;
    head_in_conv.labels_done = head_in_conv.done = false;
    // scalars: label sum 0, min +inf, max -inf, loss 0, l2 0
    if (!(step_init_done && backward))
        LAUNCH(this, "g_step_init", 0, 0,
               g_step_init(stream, scalars, g, backward ? (size_t)(nT + 8) : 0, extra_zero,
                           backward && !generic ? extra_zero_n : 0));
    step_init_done = false;
    if (labels_done) {
        // the forward pass of this train step already ran the label statistics (and maybe the head, in its last conv's epilogue)
    } else if (generic || !fast_label_stats(this, npix, y_dev))
        LAUNCH(this, "g_label_stats", 4.0 * npix, (double)npix, g_label_stats(stream, npix, y_dev, scalars));
    // dlogits scale: mean over (H, W), then mean over the (rank-local) batch.  Under data parallel every rank uses its
    // local mean; the cross-rank 1/world is applied to the all-reduced gradient in the optimizer step.
    float gscale = (float)(1.0 / ((double)outH * outW * B));
    bool head_done = false;
    if (head_was_in_conv) {
        head_done = true;
    } else if (backward && head_deferred) {
        Op& ho = ops.back();
        double fb = 4.0 * nelem(B, ho.inA.d);
        if (!fast_head_train(this, B, ho, y_dev, cfg, gscale, 2 * fb + 4.0 * npix)) {
            set_error("internal: deferred head has no fused kernel");
            return DNNCA_ESTATE;
        }
        head_done = true;
    } else {
        LAUNCH(this, "g_loss", 4.0 * npix * (backward ? 4 : 3), 20.0 * npix,
               g_loss(stream, npix, logits, y_dev, cfg, (double)npix, scalars, backward ? dlogits : nullptr, prob, gscale));
    }
    if (backward) {
        cur_op = nullptr;
        DN_TRY(ig_begin_backward(this));
        // bucketed gradient all-reduce (data parallel, large models): see model.h.  Not for the pixel-group plan (its slabs are
        // folded into the gradient vector by the launch that ENDS the backward pass; 34.7 KB anyway) and not with an L2
        // regulariser (its gradient is added after the loop).
        bucketing = false;
        collectives_last_step = 0;
        if (comm && !dry && (size_t)nT * 4 > (1u << 20) && desc.l2 == 0.f && !head_defer_ok) {
            if (bucket_state == 0) {
                int64_t prev = -1;
                bucket_state = 1;
                for (const Op& q : ops) {
                    int64_t lo = q.w_off >= 0 ? q.w_off : q.b_off;
                    if (q.b_off >= 0 && q.b_off < lo) lo = q.b_off;
                    if (lo < 0) continue;
                    if (lo <= prev) { bucket_state = -1; break; }
                    prev = lo;
                }
                if (const char* e = getenv("DNNCA_BUCKET_BYTES")) bucket_bytes = (size_t)atol(e) > 4096 ? (size_t)atol(e) : 4096;
            }
            bucketing = bucket_state == 1;
            bucket_hi = bucket_fin = nT;
        }
        const std::vector<int> order = pass_order(true);          // (lockstep over the mulmo encoders unless this pass sends gradient buckets)
        op_done.assign(ops.size(), 0);
        for (size_t ok_ = 0; ok_ < order.size(); ++ok_) {
            const int i = order[ok_];
            if (op_done[i]) continue;
            Op& o = ops[i];
            cur_op = &o.name;
            if (bucketing && i + 1 < (int)ops.size()) {
                // everything from the previous (later) op's parameters to the end of the vector is final now
                const Op& d = ops[i + 1];
                int64_t lo = d.w_off >= 0 ? d.w_off : d.b_off;
                if (d.b_off >= 0 && d.b_off < lo) lo = d.b_off;
                if (lo >= 0 && lo < bucket_fin) bucket_fin = lo;
                if ((size_t)(bucket_hi - bucket_fin) * 4 >= bucket_bytes) {
                    DN_TRY(send_bucket(bucket_fin, bucket_hi));
                    bucket_hi = bucket_fin;
                }
            }
            switch (o.type) {
                case OP_HEAD: {
                    if (head_done) break;
                    if (o.maskA) { set_error("internal: masked head gradient needs the fused kernel"); return DNNCA_ESTATE; }
                    double fb = 4.0 * nelem(B, o.inA.d);
                    LAUNCH(this, "g_head_bwd", 2 * fb + 4.0 * npix, 4.0 * nelem(B, o.inA.d),
                           g_head_bwd(stream, B, o.inA.d, p + o.w_off, dlogits, o.inA.g, g + o.w_off, g + o.b_off));
                    break;
                }
                case OP_CONV: {
                    int Cin = o.inA.d.C + o.inB.d.C;
                    double ob = 4.0 * nelem(B, o.out.d), ib = 4.0 * (nelem(B, o.inA.d) + nelem(B, o.inB.d));
                    double flops = 2.0 * B * o.out.d.H * o.out.d.W * o.k * o.k * Cin * o.out.d.C;
                    if (first_done == &o) {                         // its weight gradient rode in the previous launch (k_first3)
                        first_done = nullptr;
                        break;
                    }
                    if (tail_done == &o && head_was_in_conv) {      // its backward ran with the head, inside the forward pass (fast_tail3)
                        tail_done = nullptr;
                        break;
                    }
                    if (!generic && i >= 2 && fused_up_bwd(this, B, (size_t)i)) {       // second conv, first conv, transposed conv of a decoder block in one launch
                        op_done[i - 1] = op_done[i - 2] = 1;
                        break;
                    }
                    if (!generic && (fast_conv_bwd(this, B, o, ob, ib, flops) || fast_first_conv_bwd(this, B, o, ob, ib, flops) ||
                                     ig_conv_bwd(this, B, o, ob, ib, flops))) break;
                    if (!all_f32(o)) return DNNCA_ESTATE;
                    if (o.maskA || o.maskB) { set_error("internal: masked conv gradient has no tuned kernel"); return DNNCA_ESTATE; }
                    if (o.alpha >= 0.f && !o.premasked)
                        LAUNCH(this, "g_act_bwd", 3 * ob, ob / 4,
                               g_act_bwd(stream, (size_t)nelem(B, o.out.d), o.out.g.p, o.out.d.p, o.alpha));
                    LAUNCH(this, "g_conv_wgrad", ob + ib, flops,
                           g_conv_wgrad(stream, B, o.inA.d, o.inB.d, o.out.g, g + o.w_off, g + o.b_off, o.k));
                    if (o.need_din)
                        LAUNCH(this, "g_conv_dgrad", ob + ib, flops,
                               g_conv_dgrad(stream, B, o.out.g, p + o.w_off, o.inA.g, o.accA, o.inB.g, o.accB, o.k));
                    break;
                }
                case OP_BN: {
                    double tb = 4.0 * nelem(B, o.inA.d);
                    double n = (double)B * o.inA.d.H * o.inA.d.W;
                    if (!generic && fast_bn_bwd(this, B, o)) break;
                    if (!all_f32(o)) return DNNCA_ESTATE;
                    if (o.maskA) { set_error("internal: masked batch-norm gradient has no tuned kernel"); return DNNCA_ESTATE; }
                    LAUNCH(this, "g_bn_bwd_reduce", 2 * tb, tb,
                           g_bn_bwd_reduce(stream, B, o.inA.d, o.out.g, o.coef, g + o.w_off, g + o.b_off));
                    LAUNCH(this, "g_bn_bwd_apply", 3 * tb, 2 * tb,
                           g_bn_bwd_apply(stream, B, o.inA.d, o.out.g, o.inA.g, o.accA, o.coef, p + o.w_off, g + o.w_off,
                                          g + o.b_off, n));
                    break;
                }
                case OP_POOL: {
                    double bytes = 4.0 * (2 * nelem(B, o.inA.d) + 2 * nelem(B, o.out.d));
                    if (!generic && i >= 2 && fused_down_bwd(this, B, (size_t)i)) {     // pool, second conv, first conv of an encoder block in one launch
                        op_done[i - 1] = op_done[i - 2] = 1;
                        break;
                    }
                    if (!generic && i > 0 && fast_pool_into_bn(this, o, ops[i - 1])) break;   // rides in the backward passes of the BatchNorm in front of it
                    if (!generic && i > 0 && fast_pool_fold(this, o, ops[i - 1])) break;      // rides in the next launch (the conv's backward)
                    if (!generic && fast_pool_bwd(this, B, o, bytes)) break;
                    if (!all_f32(o)) return DNNCA_ESTATE;
                    if (o.maskA) { set_error("internal: masked pool gradient has no tuned kernel"); return DNNCA_ESTATE; }
                    LAUNCH(this, "g_pool_bwd", bytes, 0,
                           g_pool_bwd(stream, B, o.inA.d, o.out.d, o.out.g, o.inA.g, o.accA, o.k));
                    break;
                }
                case OP_TCONV: {
                    double ob = 4.0 * nelem(B, o.out.d), ib = 4.0 * nelem(B, o.inA.d);
                    double flops = 2.0 * nelem(B, o.out.d) * o.inA.d.C;
                    if (tconv_done == &o) {                         // its backward rode in the previous launch (k_pgbwd TCF)
                        tconv_done = nullptr;
                        break;
                    }
                    if (!generic && (fast_tconv_bwd(this, B, o, ob, ib, flops) || ig_tconv_bwd(this, B, o, ob, ib, flops))) break;
                    if (!all_f32(o)) return DNNCA_ESTATE;
                    if (o.maskA) { set_error("internal: masked transposed-conv gradient has no tuned kernel"); return DNNCA_ESTATE; }
                    LAUNCH(this, "g_tconv_wgrad", ob + ib, flops,
                           g_tconv_wgrad(stream, B, o.inA.d, o.out.g, g + o.w_off, g + o.b_off, o.k));
                    LAUNCH(this, "g_tconv_dgrad", ob + ib, flops,
                           g_tconv_dgrad(stream, B, o.out.g, p + o.w_off, o.inA.g, o.accA, o.k));
                    break;
                }
            }
        }
        DN_TRY(ig_finish_wgrad(this));
        DN_TRY(wg_side_join());
        DN_TRY(fast_finish_backward(this));
        if (desc.l2 > 0.f) {
            for (auto& pi : params) {
                if (!pi.trainable || pi.name.size() < 7 || pi.name.rfind(".kernel") != pi.name.size() - 7) continue;
                LAUNCH(this, "g_l2", 12.0 * pi.size, 4.0 * pi.size,
                       g_l2(stream, (size_t)pi.size, p + pi.offset, g + pi.offset, desc.l2, scalars));
            }
        }
    } else if (desc.l2 > 0.f) {
        // evaluation: Keras adds the regulariser to the reported loss as well; it needs no gradient here.
        // (g is scratch in this mode.)
        for (auto& pi : params) {
            if (!pi.trainable || pi.name.size() < 7 || pi.name.rfind(".kernel") != pi.name.size() - 7) continue;
            LAUNCH(this, "g_l2", 12.0 * pi.size, 4.0 * pi.size,
                   g_l2(stream, (size_t)pi.size, p + pi.offset, g + pi.offset, desc.l2, scalars));
        }
    }
    if (backward && !comm && !dry && merged_launches()) {
        // optimizer_step() follows every backward pass: its Adam launch also writes the step outputs (one launch fewer)
        fin_pending.on = true;
        fin_pending.cfg = cfg;
        fin_pending.n_label = (double)npix;
        fin_pending.inv_batch_hw = 1.0 / ((double)outH * outW * B);
        return DNNCA_OK;
    }
    LAUNCH(this, "g_finalize_scalars", 0, 0,
           g_finalize_scalars(stream, scalars, cfg, (double)npix, 1.0 / ((double)outH * outW * B), out5));
    return DNNCA_OK;
}

// [lo, hi) of the flat gradient vector is final on `stream`: sum it over the ranks on the communication stream
int Model::send_bucket(int64_t lo, int64_t hi) {
    if (hi <= lo) return DNNCA_OK;
    if (!comm_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ev_bucket, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_comm_done, hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(ev_bucket, stream));
    HIP_TRY(hipStreamWaitEvent(comm_stream, ev_bucket, 0));
    if (wg_pending) {          // weight gradients of the bucket's layers may still be running on the side stream (FIFO: one event covers them)
        HIP_TRY(hipEventRecord(wg_bucket, wg_stream));
        HIP_TRY(hipStreamWaitEvent(comm_stream, wg_bucket, 0));
    }
    ncclResult_t r = ncclAllReduce(g + lo, g + lo, (size_t)(hi - lo), ncclFloat, ncclSum, comm, comm_stream);
    if (r != ncclSuccess) { set_error("ncclAllReduce(bucket): %s", ncclGetErrorString(r)); return DNNCA_ECOMM; }
    ++collectives_last_step;
    return DNNCA_OK;
}

int Model::optimizer_step(float lr) {
    cur_op = nullptr;
    float gscale = 1.0f;
    if (comm && !dry && bucketing) {
        // the backward pass has sent the finalised suffix in buckets; what is left -- the first layers and the step's loss in the
        // tail -- follows on the same (communication) stream, and Adam waits for all of it
        bucketing = false;
        DN_TRY(send_bucket(0, bucket_hi));
        DN_TRY(send_bucket(nT, nT + 1));
        HIP_TRY(hipEventRecord(ev_comm_done, comm_stream));
        HIP_TRY(hipStreamWaitEvent(stream, ev_comm_done, 0));
        gscale = 1.0f / (float)world;
    } else if (comm && !dry) {
        // one all-reduce over [gradients ..., loss]: MirroredStrategy's cross-replica sum (engine.py:262) [TF-2.6]
        ncclResult_t r = ncclAllReduce(g, g, (size_t)nT + 1, ncclFloat, ncclSum, comm, stream);
        if (r != ncclSuccess) { set_error("ncclAllReduce: %s", ncclGetErrorString(r)); return DNNCA_ECOMM; }
        collectives_last_step = 1;
        gscale = 1.0f / (float)world;
    }
    iterations += dry ? 0 : 1;
    double t = (double)(dry ? 1 : iterations);
    float lr_t = (float)((double)lr * std::sqrt(1.0 - std::pow((double)beta2, t)) / (1.0 - std::pow((double)beta1, t)));
    if (fin_pending.on && fold_deferred) {
        fin_pending.on = false;
        return fast_fold_adam(this, lr_t, fin_pending.cfg, fin_pending.n_label, fin_pending.inv_batch_hw);
    }
    if (fold_deferred) { set_error("internal: deferred gradient fold without a single-replica optimizer step"); return DNNCA_ESTATE; }
    if (fin_pending.on) {
        fin_pending.on = false;
        LAUNCH(this, "g_adam", 28.0 * nT, 10.0 * nT,
               g_adam_finalize(stream, (size_t)nT, p, g, m, v, lr_t, beta1, beta2, eps, gscale, scalars, fin_pending.cfg,
                               fin_pending.n_label, fin_pending.inv_batch_hw, out5));
        return DNNCA_OK;
    }
    LAUNCH(this, "g_adam", 28.0 * nT, 10.0 * nT, g_adam(stream, (size_t)nT, p, g, m, v, lr_t, beta1, beta2, eps, gscale));
    return DNNCA_OK;
}

}  // namespace dnnca

// =================================================================================================== C ABI
using namespace dnnca;

#define MODEL(h)                                             \
    Model* M = reinterpret_cast<Model*>(h);                  \
    if (!M) { set_error("null model handle"); return DNNCA_EINVAL; }

extern "C" {

const char* dnnca_version(void) { return "dnnca 0.1 (gfx950)"; }
const char* dnnca_last_error(void) { return g_err; }

int dnnca_device_count(int* count) {
    if (!count) return DNNCA_EINVAL;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return DNNCA_EHIP; }
    *count = n;
    return DNNCA_OK;
}

int dnnca_init(int device_ordinal) {
    HIP_TRY(hipSetDevice(device_ordinal));
    return DNNCA_OK;
}

int dnnca_model_create(const dnnca_model_desc* desc, void** model_out) {
    if (!desc || !model_out) { set_error("null argument"); return DNNCA_EINVAL; }
    *model_out = nullptr;
    Model* M = new Model();
    M->desc = *desc;
    if (M->desc.n_conv == 0) M->desc.n_conv = 2;
    hipError_t e = hipGetDevice(&M->device);
    if (e != hipSuccess) { set_error("no HIP device: %s", hipGetErrorString(e)); delete M; return DNNCA_EHIP; }
    e = hipStreamCreateWithFlags(&M->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { set_error("hipStreamCreate: %s", hipGetErrorString(e)); M->stream = nullptr; delete M; return DNNCA_EHIP; }
    int rc = M->build();
    if (rc != DNNCA_OK) { delete M; return rc; }
    *model_out = M;
    return DNNCA_OK;
}

int dnnca_model_destroy(void* model) {
    MODEL(model);
    (void)hipStreamSynchronize(M->stream);
    delete M;
    return DNNCA_OK;
}

int dnnca_param_count(void* model, int* count) {
    MODEL(model);
    *count = (int)M->params.size();
    return DNNCA_OK;
}

int dnnca_param_info(void* model, int index, char* name, size_t name_cap, int64_t shape[4], int* ndim, int* trainable,
                     int64_t* offset) {
    MODEL(model);
    if (index < 0 || index >= (int)M->params.size()) { set_error("param index out of range"); return DNNCA_EINVAL; }
    const ParamInfo& pi = M->params[index];
    if (name && name_cap) snprintf(name, name_cap, "%s", pi.name.c_str());
    if (shape) for (int i = 0; i < 4; ++i) shape[i] = pi.shape[i];
    if (ndim) *ndim = pi.ndim;
    if (trainable) *trainable = pi.trainable;
    if (offset) *offset = pi.offset;
    return DNNCA_OK;
}

int dnnca_num_trainable(void* model, int64_t* n) { MODEL(model); *n = M->nT; return DNNCA_OK; }
int dnnca_num_state(void* model, int64_t* n) { MODEL(model); *n = M->nS; return DNNCA_OK; }

static int copy_flat(Model* M, float* dev, int64_t have, const float* src, float* dst, int64_t n) {
    if (n != have) { set_error("flat vector length %lld != %lld", (long long)n, (long long)have); return DNNCA_EINVAL; }
    if (n == 0) return DNNCA_OK;
    if (src) HIP_TRY(hipMemcpyAsync(dev, src, (size_t)n * 4, hipMemcpyHostToDevice, M->stream));
    if (dst) HIP_TRY(hipMemcpyAsync(dst, dev, (size_t)n * 4, hipMemcpyDeviceToHost, M->stream));
    HIP_TRY(hipStreamSynchronize(M->stream));
    return DNNCA_OK;
}

int dnnca_set_params(void* model, const float* flat, int64_t n) { MODEL(model); return copy_flat(M, M->p, M->nT, flat, nullptr, n); }
int dnnca_get_params(void* model, float* flat, int64_t n) { MODEL(model); return copy_flat(M, M->p, M->nT, nullptr, flat, n); }
int dnnca_set_state(void* model, const float* flat, int64_t n) { MODEL(model); return copy_flat(M, M->state, M->nS, flat, nullptr, n); }
int dnnca_get_state(void* model, float* flat, int64_t n) { MODEL(model); return copy_flat(M, M->state, M->nS, nullptr, flat, n); }
int dnnca_get_grads(void* model, float* flat, int64_t n) { MODEL(model); return copy_flat(M, M->g, M->nT, nullptr, flat, n); }

int dnnca_set_opt_state(void* model, const float* m, const float* v, int64_t n, int64_t iterations) {
    MODEL(model);
    DN_TRY(copy_flat(M, M->m, M->nT, m, nullptr, n));
    DN_TRY(copy_flat(M, M->v, M->nT, v, nullptr, n));
    M->iterations = iterations;
    return DNNCA_OK;
}

int dnnca_get_opt_state(void* model, float* m, float* v, int64_t n, int64_t* iterations) {
    MODEL(model);
    DN_TRY(copy_flat(M, M->m, M->nT, nullptr, m, n));
    DN_TRY(copy_flat(M, M->v, M->nT, nullptr, v, n));
    if (iterations) *iterations = M->iterations;
    return DNNCA_OK;
}

int dnnca_set_adam(void* model, float beta1, float beta2, float epsilon) {
    MODEL(model);
    M->beta1 = beta1;
    M->beta2 = beta2;
    M->eps = epsilon;
    return DNNCA_OK;
}

static int convert_out(Model* M, const float* h, dnnca_step_out* out);
static int confusion_begin(Model* M, const float* thresholds, int n, std::vector<int>& order);
static int confusion_finish(Model* M, int n, const std::vector<int>& order, dnnca_confusion* out);

static int read_out(Model* M, dnnca_step_out* out) {
    float h[5];
    HIP_TRY(hipMemcpyAsync(h, M->out5, sizeof(h), hipMemcpyDeviceToHost, M->stream));
    HIP_TRY(hipStreamSynchronize(M->stream));
    DN_TRY(M->flush_profile());
    return convert_out(M, h, out);
}

static int convert_out(Model* M, const float* h, dnnca_step_out* out) {
    out->loss = h[0] / (float)M->world;   // after the all-reduce the slot holds the sum over ranks of the local means
    out->positive_rate = h[1];
    out->weight = h[2];
    out->label_min = h[3];
    out->label_max = h[4];
    // utils/losses.py:91-92 assert_on_max / assert_on_min, :30 assert_on_weight
    if (h[4] > 1.0f || h[3] < 0.0f) { set_error("label outside [0, 1]: min %g max %g (assert_on_min/assert_on_max)", h[3], h[4]); return DNNCA_EASSERT; }
    if (h[2] < 0.0f) { set_error("negative class weight %g (assert_on_weight)", h[2]); return DNNCA_EASSERT; }
    return DNNCA_OK;
}

int dnnca_forward_dev(void* model, const float* x_dev, int batch, int training) {
    MODEL(model);
    DN_TRY(M->forward(x_dev, batch, training != 0));
    size_t npix = (size_t)batch * M->outH * M->outW;
    LAUNCH(M, "g_sigmoid", 8.0 * npix, 4.0 * npix, g_sigmoid(M->stream, npix, M->logits, M->prob));
    return DNNCA_OK;
}

int dnnca_forward(void* model, const float* x_nhwc, int batch, int training, float* prob_out, float* logit_out) {
    MODEL(model);
    if (batch < 1 || batch > M->desc.max_batch) { set_error("batch %d outside [1, %d]", batch, M->desc.max_batch); return DNNCA_EINVAL; }
    size_t nx = (size_t)batch * M->desc.height * M->desc.width * M->desc.in_channels;
    size_t npix = (size_t)batch * M->outH * M->outW;
    HIP_TRY(hipMemcpyAsync(M->x_stage, x_nhwc, nx * 4, hipMemcpyHostToDevice, M->stream));
    DN_TRY(dnnca_forward_dev(model, M->x_stage, batch, training));
    if (prob_out) HIP_TRY(hipMemcpyAsync(prob_out, M->prob, npix * 4, hipMemcpyDeviceToHost, M->stream));
    if (logit_out) HIP_TRY(hipMemcpyAsync(logit_out, M->logits, npix * 4, hipMemcpyDeviceToHost, M->stream));
    HIP_TRY(hipStreamSynchronize(M->stream));
    return M->flush_profile();
}

int dnnca_train_step_dev(void* model, const float* x_dev, const float* y_dev, int batch, float lr, const dnnca_loss_cfg* cfg,
                         dnnca_step_out* out) {
    MODEL(model);
    if (!cfg) { set_error("null loss cfg"); return DNNCA_EINVAL; }
    M->defer_head = true;
    // (label smoothing blurs the labels in loss_and_backward first: then the head cannot run inside the forward pass)
    M->head_in_conv.requested = !cfg->label_smoothing && M->merged_launches() && !getenv("DNNCA_NO_HEAD_IN_CONV");
    M->head_in_conv.y = y_dev;
    M->head_in_conv.cfg = *cfg;
    int frc = M->forward(x_dev, batch, true);
    M->defer_head = false;
    M->head_in_conv.requested = false;
    DN_TRY(frc);
    DN_TRY(M->loss_and_backward(y_dev, batch, *cfg, true));
    DN_TRY(M->optimizer_step(lr));
    if (out) return read_out(M, out);
    return DNNCA_OK;
}

int dnnca_train_step(void* model, const float* x_nhwc, const float* y_hw, int batch, float lr, const dnnca_loss_cfg* cfg,
                     dnnca_step_out* out) {
    MODEL(model);
    if (batch < 1 || batch > M->desc.max_batch) { set_error("batch %d outside [1, %d]", batch, M->desc.max_batch); return DNNCA_EINVAL; }
    size_t nx = (size_t)batch * M->desc.height * M->desc.width * M->desc.in_channels;
    size_t npix = (size_t)batch * M->outH * M->outW;
    HIP_TRY(hipMemcpyAsync(M->x_stage, x_nhwc, nx * 4, hipMemcpyHostToDevice, M->stream));
    HIP_TRY(hipMemcpyAsync(M->y_stage, y_hw, npix * 4, hipMemcpyHostToDevice, M->stream));
    dnnca_step_out tmp;
    return dnnca_train_step_dev(model, M->x_stage, M->y_stage, batch, lr, cfg, out ? out : &tmp);
}

int dnnca_eval_step(void* model, const float* x_nhwc, const float* y_hw, int batch, const dnnca_loss_cfg* cfg,
                    dnnca_step_out* out, float* prob_out) {
    MODEL(model);
    if (!cfg) { set_error("null loss cfg"); return DNNCA_EINVAL; }
    if (batch < 1 || batch > M->desc.max_batch) { set_error("batch %d outside [1, %d]", batch, M->desc.max_batch); return DNNCA_EINVAL; }
    size_t nx = (size_t)batch * M->desc.height * M->desc.width * M->desc.in_channels;
    size_t npix = (size_t)batch * M->outH * M->outW;
    HIP_TRY(hipMemcpyAsync(M->x_stage, x_nhwc, nx * 4, hipMemcpyHostToDevice, M->stream));
    HIP_TRY(hipMemcpyAsync(M->y_stage, y_hw, npix * 4, hipMemcpyHostToDevice, M->stream));
    DN_TRY(M->forward(M->x_stage, batch, false));
    DN_TRY(M->loss_and_backward(M->y_stage, batch, *cfg, false));
    if (prob_out) HIP_TRY(hipMemcpyAsync(prob_out, M->prob, npix * 4, hipMemcpyDeviceToHost, M->stream));
    dnnca_step_out tmp;
    int world = M->world;
    M->world = 1;   // evaluation is rank-local: the loss slot was not all-reduced
    int rc = read_out(M, out ? out : &tmp);
    M->world = world;
    return rc;
}

int dnnca_last_step_out(void* model, dnnca_step_out* out) {
    MODEL(model);
    if (!out) return DNNCA_EINVAL;
    return read_out(M, out);
}

// ---- input pipeline: staging ring + copy stream (model.h) ------------------------------------------------------------------
static size_t stage_align(size_t b) { return (b + 255) & ~(size_t)255; }

int dnnca_stage_init(void* model, int slots, size_t bytes_per_slot) {
    MODEL(model);
    if (slots < 1 || slots > Model::kStageSlots) { set_error("staging slots %d outside [1, %d]", slots, Model::kStageSlots); return DNNCA_EINVAL; }
    if (M->stage_slots) { set_error("the staging ring exists already"); return DNNCA_ESTATE; }
    const size_t xb = (size_t)M->desc.max_batch * M->desc.height * M->desc.width * M->desc.in_channels * 4;
    const size_t yb = (size_t)M->desc.max_batch * M->outH * M->outW * 4;
    if (bytes_per_slot < stage_align(xb) + yb) bytes_per_slot = stage_align(xb) + yb;      // at least one float batch (x, y)
    HIP_TRY(hipStreamCreateWithFlags(&M->copy_stream, hipStreamNonBlocking));
    HIP_TRY(hipHostMalloc((void**)&M->out_ring, (size_t)Model::kStageSlots * 8 * sizeof(float), hipHostMallocDefault));
    for (int i = 0; i < slots; ++i) {
        Model::StageSlot& sl = M->stage[i];
        DN_TRY(M->alloc((void**)&sl.x, stage_align(bytes_per_slot)));
        HIP_TRY(hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    M->stage_bytes = stage_align(bytes_per_slot);
    M->stage_slots = slots;
    HIP_TRY(hipStreamSynchronize(M->stream));          // the allocations' memsets: nothing of them is left for the copy stream to race
    return DNNCA_OK;
}

// May be called from other host threads (each working on slots of its own) while the owning thread enqueues steps: it touches
// only the slot it was handed, the copy stream and the HIP runtime.  With pageable host memory the call returns when the copy is done;
// the GPU keeps working on the main stream meanwhile.
int dnnca_stage_upload(void* model, int slot, const void* host_a, size_t bytes_a, const void* host_b, size_t bytes_b,
                       void** a_dev, void** b_dev) {
    MODEL(model);
    if (slot < 0 || slot >= M->stage_slots) { set_error("staging slot %d outside [0, %d)", slot, M->stage_slots); return DNNCA_EINVAL; }
    if (stage_align(bytes_a) + bytes_b > M->stage_bytes) {
        set_error("staging slot holds %zu bytes, asked for %zu + %zu", M->stage_bytes, bytes_a, bytes_b);
        return DNNCA_EINVAL;
    }
    HIP_TRY(hipSetDevice(M->device));                  // the current device is per host thread
    Model::StageSlot& sl = M->stage[slot];
    char* base = reinterpret_cast<char*>(sl.x);
    if (sl.has_done) HIP_TRY(hipStreamWaitEvent(M->copy_stream, sl.done, 0));      // the step that read the slot's last batch
    if (host_a && bytes_a) HIP_TRY(hipMemcpyAsync(base, host_a, bytes_a, hipMemcpyHostToDevice, M->copy_stream));
    if (host_b && bytes_b) HIP_TRY(hipMemcpyAsync(base + stage_align(bytes_a), host_b, bytes_b, hipMemcpyHostToDevice, M->copy_stream));
    HIP_TRY(hipEventRecord(sl.uploaded, M->copy_stream));
    if (a_dev) *a_dev = base;
    if (b_dev) *b_dev = base + stage_align(bytes_a);
    return DNNCA_OK;
}

int dnnca_stage_uploaded(void* model, int slot) {
    MODEL(model);
    if (slot < 0 || slot >= M->stage_slots) { set_error("staging slot %d outside [0, %d)", slot, M->stage_slots); return DNNCA_EINVAL; }
    HIP_TRY(hipEventSynchronize(M->stage[slot].uploaded));
    return DNNCA_OK;
}

int dnnca_stage_wait(void* model, int slot) {
    MODEL(model);
    if (slot < 0 || slot >= M->stage_slots) { set_error("staging slot %d outside [0, %d)", slot, M->stage_slots); return DNNCA_EINVAL; }
    HIP_TRY(hipStreamWaitEvent(M->stream, M->stage[slot].uploaded, 0));
    return DNNCA_OK;
}

int dnnca_train_step_staged(void* model, int slot, const float* x_dev, const float* y_dev, int batch, float lr,
                            const dnnca_loss_cfg* cfg) {
    MODEL(model);
    if (slot < 0 || slot >= M->stage_slots) { set_error("staging slot %d outside [0, %d)", slot, M->stage_slots); return DNNCA_EINVAL; }
    if (batch < 1 || batch > M->desc.max_batch) { set_error("batch %d outside [1, %d]", batch, M->desc.max_batch); return DNNCA_EINVAL; }
    Model::StageSlot& sl = M->stage[slot];
    HIP_TRY(hipStreamWaitEvent(M->stream, sl.uploaded, 0));
    DN_TRY(dnnca_train_step_dev(model, x_dev, y_dev, batch, lr, cfg, nullptr));
    HIP_TRY(hipMemcpyAsync(M->out_ring + slot * 8, M->out5, 5 * sizeof(float), hipMemcpyDeviceToHost, M->stream));
    HIP_TRY(hipEventRecord(sl.done, M->stream));
    sl.has_done = true;
    sl.is_eval = false;
    sl.batch = batch;
    return DNNCA_OK;
}

int dnnca_staged_out(void* model, int slot, dnnca_step_out* out) {
    MODEL(model);
    if (!out) return DNNCA_EINVAL;
    if (slot < 0 || slot >= M->stage_slots || !M->stage[slot].has_done) { set_error("staging slot %d has run no step", slot); return DNNCA_ESTATE; }
    HIP_TRY(hipEventSynchronize(M->stage[slot].done));
    if (M->prof_mode) DN_TRY(M->flush_profile());
    const int world = M->world;
    if (M->stage[slot].is_eval) M->world = 1;      // evaluation is rank-local: the loss slot was not all-reduced
    const int rc = convert_out(M, M->out_ring + slot * 8, out);
    M->world = world;
    return rc;
}

// ---- staged evaluation: keras Model.evaluate (engine.py:198-203) over the staging ring ---------------------------------
int dnnca_eval_begin(void* model, const float* thresholds, int n) {
    MODEL(model);
    if (n < 0 || n > DNNCA_CONF_MAX_THR || (n > 0 && !thresholds)) { set_error("bad thresholds (0..%d)", DNNCA_CONF_MAX_THR); return DNNCA_EINVAL; }
    if (!M->stage_slots) { set_error("dnnca_eval_begin before dnnca_stage_init"); return DNNCA_ESTATE; }
    M->eval_order.clear();
    if (n > 0) DN_TRY(confusion_begin(M, thresholds, n, M->eval_order));
    M->eval_active = true;
    return DNNCA_OK;
}

int dnnca_eval_step_staged(void* model, int slot, const float* x_dev, const float* y_dev, int batch, const dnnca_loss_cfg* cfg) {
    MODEL(model);
    if (!M->eval_active) { set_error("dnnca_eval_step_staged outside dnnca_eval_begin .. dnnca_eval_end"); return DNNCA_ESTATE; }
    if (!cfg) { set_error("null loss cfg"); return DNNCA_EINVAL; }
    if (slot < 0 || slot >= M->stage_slots) { set_error("staging slot %d outside [0, %d)", slot, M->stage_slots); return DNNCA_EINVAL; }
    if (batch < 1 || batch > M->desc.max_batch) { set_error("batch %d outside [1, %d]", batch, M->desc.max_batch); return DNNCA_EINVAL; }
    Model::StageSlot& sl = M->stage[slot];
    HIP_TRY(hipStreamWaitEvent(M->stream, sl.uploaded, 0));
    DN_TRY(M->forward(x_dev, batch, false));
    DN_TRY(M->loss_and_backward(y_dev, batch, *cfg, false));
    const int n = (int)M->eval_order.size();
    if (n > 0)      // the metrics see the labels as given (a smoothed copy exists only inside the loss, utils/losses.py:62-67)
        g_confusion_hist(M->stream, (size_t)batch * M->outH * M->outW, M->prob, y_dev, M->thr_dev, n,
                         reinterpret_cast<unsigned long long*>(M->conf_dev));
    HIP_TRY(hipMemcpyAsync(M->out_ring + slot * 8, M->out5, 5 * sizeof(float), hipMemcpyDeviceToHost, M->stream));
    HIP_TRY(hipEventRecord(sl.done, M->stream));
    sl.has_done = true;
    sl.is_eval = true;
    sl.batch = batch;
    return DNNCA_OK;
}

int dnnca_eval_end(void* model, dnnca_confusion* out) {
    MODEL(model);
    if (!M->eval_active) { set_error("dnnca_eval_end without dnnca_eval_begin"); return DNNCA_ESTATE; }
    M->eval_active = false;
    const int n = (int)M->eval_order.size();
    if (n > 0 && !out) { set_error("null confusion output"); return DNNCA_EINVAL; }
    if (n > 0) return confusion_finish(M, n, M->eval_order, out);
    HIP_TRY(hipStreamSynchronize(M->stream));
    return DNNCA_OK;
}

int dnnca_sync(void* model) {
    MODEL(model);
    HIP_TRY(hipStreamSynchronize(M->stream));
    return M->flush_profile();
}

int dnnca_dev_alloc(void** dev_ptr, size_t bytes) {
    if (!dev_ptr) return DNNCA_EINVAL;
    HIP_TRY(hipMalloc(dev_ptr, bytes ? bytes : 4));
    return DNNCA_OK;
}
int dnnca_dev_free(void* dev_ptr) { HIP_TRY(hipFree(dev_ptr)); return DNNCA_OK; }
int dnnca_memcpy_h2d(void* dev_dst, const void* host_src, size_t bytes) { HIP_TRY(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice)); return DNNCA_OK; }
int dnnca_memcpy_d2h(void* host_dst, const void* dev_src, size_t bytes) { HIP_TRY(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost)); return DNNCA_OK; }

// thresholds -> ascending on the device (the histogram kernel wants them sorted), histogram zeroed; order[t] = caller's position
static int confusion_begin(Model* M, const float* thresholds, int n, std::vector<int>& order) {
    order.resize(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return thresholds[a] < thresholds[b]; });
    std::vector<float> sorted(n);
    for (int i = 0; i < n; ++i) {
        sorted[i] = thresholds[order[i]];
        if (sorted[i] != sorted[i]) { set_error("threshold %d is NaN", order[i]); return DNNCA_EINVAL; }
    }
    HIP_TRY(hipMemcpyAsync(M->thr_dev, sorted.data(), (size_t)n * 4, hipMemcpyHostToDevice, M->stream));
    HIP_TRY(hipMemsetAsync(M->conf_dev, 0, (size_t)2 * (n + 1) * 8, M->stream));
    HIP_TRY(hipStreamSynchronize(M->stream));          // `sorted` is a local
    return DNNCA_OK;
}

// histogram [positives | negatives][n + 1 bins] -> TP / FP / FN / TN per threshold, in the caller's order
static int confusion_finish(Model* M, int n, const std::vector<int>& order, dnnca_confusion* out) {
    const size_t hbytes = (size_t)2 * (n + 1) * 8;
    std::vector<unsigned long long> h((size_t)2 * (n + 1));
    HIP_TRY(hipMemcpyAsync(h.data(), M->conf_dev, hbytes, hipMemcpyDeviceToHost, M->stream));
    HIP_TRY(hipStreamSynchronize(M->stream));
    const unsigned long long* pos = h.data();
    const unsigned long long* neg = h.data() + (n + 1);
    unsigned long long P = 0, N = 0;
    for (int b = 0; b <= n; ++b) { P += pos[b]; N += neg[b]; }
    unsigned long long tp = 0, fp = 0;            // suffix sums: prob > sorted[t]  <=>  bin > t
    for (int t = n - 1; t >= 0; --t) {
        tp += pos[t + 1];
        fp += neg[t + 1];
        dnnca_confusion& o = out[order[t]];
        o.tp = (double)tp;
        o.fp = (double)fp;
        o.fn = (double)(P - tp);
        o.tn = (double)(N - fp);
    }
    return DNNCA_OK;
}

static int confusion_counts(Model* M, size_t npix, const float* thresholds, int n, dnnca_confusion* out) {
    if (M->eval_active) { set_error("dnnca_pixel_confusion* inside dnnca_eval_begin .. dnnca_eval_end (the histogram is in use)"); return DNNCA_ESTATE; }
    std::vector<int> order;
    DN_TRY(confusion_begin(M, thresholds, n, order));
    g_confusion_hist(M->stream, npix, M->prob, M->y_stage, M->thr_dev, n, reinterpret_cast<unsigned long long*>(M->conf_dev));
    return confusion_finish(M, n, order, out);
}

int dnnca_pixel_confusion(void* model, const float* y_hw, int batch, const float* thresholds, int n, dnnca_confusion* out) {
    MODEL(model);
    if (n < 1 || n > DNNCA_CONF_MAX_THR || !out || !thresholds || !y_hw) {
        set_error("bad confusion arguments (1..%d thresholds)", DNNCA_CONF_MAX_THR);
        return DNNCA_EINVAL;
    }
    if (batch < 1 || batch > M->desc.max_batch) { set_error("batch out of range"); return DNNCA_EINVAL; }
    size_t npix = (size_t)batch * M->outH * M->outW;
    HIP_TRY(hipMemcpyAsync(M->y_stage, y_hw, npix * 4, hipMemcpyHostToDevice, M->stream));
    return confusion_counts(M, npix, thresholds, n, out);
}

int dnnca_pixel_confusion_of(void* model, const float* prob_hw, const float* y_hw, int64_t n_pixels, const float* thresholds,
                             int n, dnnca_confusion* out) {
    MODEL(model);
    if (n < 1 || n > DNNCA_CONF_MAX_THR || !out || !thresholds || !y_hw || !prob_hw) {
        set_error("bad confusion arguments (1..%d thresholds)", DNNCA_CONF_MAX_THR);
        return DNNCA_EINVAL;
    }
    int64_t cap = (int64_t)M->desc.max_batch * M->outH * M->outW;
    if (n_pixels < 1 || n_pixels > cap) { set_error("n_pixels %lld outside 1..%lld", (long long)n_pixels, (long long)cap); return DNNCA_EINVAL; }
    HIP_TRY(hipMemcpyAsync(M->prob, prob_hw, (size_t)n_pixels * 4, hipMemcpyHostToDevice, M->stream));
    HIP_TRY(hipMemcpyAsync(M->y_stage, y_hw, (size_t)n_pixels * 4, hipMemcpyHostToDevice, M->stream));
    return confusion_counts(M, (size_t)n_pixels, thresholds, n, out);
}

// ------------------------------------------------------------------------------------------------- data parallel
int dnnca_comm_unique_id(void* id_out) {
    if (!id_out) return DNNCA_EINVAL;
    static_assert(sizeof(ncclUniqueId) <= DNNCA_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) { set_error("ncclGetUniqueId: %s", ncclGetErrorString(r)); return DNNCA_ECOMM; }
    memset(id_out, 0, DNNCA_UNIQUE_ID_BYTES);
    memcpy(id_out, &id, sizeof(id));
    return DNNCA_OK;
}

int dnnca_comm_init(void* model, int rank, int world, const void* unique_id, size_t id_len) {
    MODEL(model);
    if (world < 1 || rank < 0 || rank >= world) { set_error("bad rank/world %d/%d", rank, world); return DNNCA_EINVAL; }
    M->rank = rank;
    M->world = world;
    // DNNCA_FORCE_RCCL: build a one-rank communicator too, so every collective of the DP path can be exercised on one GPU
    if (world == 1 && !(getenv("DNNCA_FORCE_RCCL") && unique_id)) return DNNCA_OK;
    if (!unique_id || id_len < sizeof(ncclUniqueId)) { set_error("unique id too short"); return DNNCA_EINVAL; }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&M->comm, world, id, rank);
    if (r != ncclSuccess) { set_error("ncclCommInitRank: %s", ncclGetErrorString(r)); M->comm = nullptr; M->world = 1; return DNNCA_ECOMM; }
    return DNNCA_OK;
}

int dnnca_comm_world(void* model, int* rank, int* world) {
    MODEL(model);
    if (rank) *rank = M->rank;
    if (world) *world = M->world;
    return DNNCA_OK;
}

int dnnca_comm_collectives(void* model, int* count) {
    MODEL(model);
    if (!count) return DNNCA_EINVAL;
    *count = M->collectives_last_step;
    return DNNCA_OK;
}

int dnnca_comm_broadcast_weights(void* model, int root) {
    MODEL(model);
    if (!M->comm) return DNNCA_OK;
    struct { float* p; int64_t n; } bufs[4] = {{M->p, M->nT}, {M->state, M->nS}, {M->m, M->nT}, {M->v, M->nT}};
    for (auto& b : bufs) {
        if (b.n == 0) continue;
        ncclResult_t r = ncclBroadcast(b.p, b.p, (size_t)b.n, ncclFloat, root, M->comm, M->stream);
        if (r != ncclSuccess) { set_error("ncclBroadcast: %s", ncclGetErrorString(r)); return DNNCA_ECOMM; }
    }
    HIP_TRY(hipStreamSynchronize(M->stream));
    return DNNCA_OK;
}

int dnnca_comm_average_state(void* model) {
    MODEL(model);
    if (!M->comm || M->nS == 0) return DNNCA_OK;
    ncclResult_t r = ncclAllReduce(M->state, M->state, (size_t)M->nS, ncclFloat, ncclSum, M->comm, M->stream);
    if (r != ncclSuccess) { set_error("ncclAllReduce(state): %s", ncclGetErrorString(r)); return DNNCA_ECOMM; }
    g_scale(M->stream, (size_t)M->nS, M->state, 1.0f / (float)M->world);
    HIP_TRY(hipStreamSynchronize(M->stream));
    return DNNCA_OK;
}

int dnnca_comm_allreduce_host(void* model, double* values, int64_t n, int op) {
    MODEL(model);
    if (n < 1 || !values) { set_error("dnnca_comm_allreduce_host: bad arguments (n = %lld)", (long long)n); return DNNCA_EINVAL; }
    if (op != 0 && op != 1) { set_error("dnnca_comm_allreduce_host: op %d (0 sum, 1 max)", op); return DNNCA_EINVAL; }
    if (!M->comm) return DNNCA_OK;
    // doubles: pixel counts of a whole validation set stay exact (float32 stops at 2^24 = 64 slices of 512 x 512);
    // chunked through the confusion staging buffer, so any length works
    double* tmp = reinterpret_cast<double*>(M->conf_dev);
    const int64_t cap = 2 * (DNNCA_CONF_MAX_THR + 1);
    for (int64_t off = 0; off < n; off += cap) {
        size_t c = (size_t)std::min<int64_t>(cap, n - off);
        HIP_TRY(hipMemcpyAsync(tmp, values + off, c * 8, hipMemcpyHostToDevice, M->stream));
        ncclResult_t r = ncclAllReduce(tmp, tmp, c, ncclDouble, op == 1 ? ncclMax : ncclSum, M->comm, M->stream);
        if (r != ncclSuccess) { set_error("ncclAllReduce(host): %s", ncclGetErrorString(r)); return DNNCA_ECOMM; }
        HIP_TRY(hipMemcpyAsync(values + off, tmp, c * 8, hipMemcpyDeviceToHost, M->stream));
        HIP_TRY(hipStreamSynchronize(M->stream));
    }
    return DNNCA_OK;
}

// ------------------------------------------------------------------------------------------------- measurement
int dnnca_timer_start(void* model) {
    MODEL(model);
    HIP_TRY(hipEventRecord(M->ev0, M->stream));
    return DNNCA_OK;
}

int dnnca_timer_stop(void* model, float* elapsed_ms) {
    MODEL(model);
    HIP_TRY(hipEventRecord(M->ev1, M->stream));
    HIP_TRY(hipEventSynchronize(M->ev1));
    if (elapsed_ms) HIP_TRY(hipEventElapsedTime(elapsed_ms, M->ev0, M->ev1));
    return M->flush_profile();
}

int dnnca_profile_enable(void* model, int mode) {
    MODEL(model);
    DN_TRY(M->flush_profile());
    M->prof_mode = mode;
    return DNNCA_OK;
}

int dnnca_profile_focus(void* model, const char* kernel_name) {
    MODEL(model);
    M->focus = kernel_name ? kernel_name : "";
    return DNNCA_OK;
}

int dnnca_profile_sample(void* model, int period) {
    MODEL(model);
    if (period < 1) { set_error("period must be >= 1"); return DNNCA_EINVAL; }
    M->prof_period = period;
    return DNNCA_OK;
}

int dnnca_profile_reset(void* model) {
    MODEL(model);
    DN_TRY(M->flush_profile());
    M->kstats.clear();
    M->kid.clear();
    return DNNCA_OK;
}

int dnnca_profile_count(void* model, int* count) {
    MODEL(model);
    DN_TRY(M->flush_profile());
    *count = (int)M->kstats.size();
    return DNNCA_OK;
}

int dnnca_profile_get(void* model, int index, char* name, size_t name_cap, int64_t* launches, double* total_ms,
                      double* algorithmic_bytes, double* flops) {
    MODEL(model);
    if (index < 0 || index >= (int)M->kstats.size()) { set_error("profile index out of range"); return DNNCA_EINVAL; }
    const KStat& k = M->kstats[index];
    if (name && name_cap) snprintf(name, name_cap, "%s", k.name.c_str());
    if (launches) *launches = k.launches;
    if (total_ms) *total_ms = k.total_ms;
    double L = k.launches ? (double)k.launches : 1.0;
    if (algorithmic_bytes) *algorithmic_bytes = k.bytes / L;
    if (flops) *flops = k.flops / L;
    return DNNCA_OK;
}

// development aid (not part of include/dnnca.h): copy the first `n` 64-bit stamps of the tuned backward kernel
int dnnca_debug_read_stamps(void* model, unsigned long long* out, int n) {
    MODEL(model);
    unsigned long long* s = fast_debug_stamps(M);
    if (!s) { set_error("no stamps"); return DNNCA_ESTATE; }
    HIP_TRY(hipStreamSynchronize(M->stream));
    HIP_TRY(hipMemcpy(out, s, (size_t)n * 8, hipMemcpyDeviceToHost));
    return DNNCA_OK;
}

// development aid (not part of include/dnnca.h): n back-to-back launches of a trivial kernel on the model's stream
__global__ void k_noop(float* p) { if (p == nullptr) *p = 0.f; }
int dnnca_debug_launch_cost(void* model, int n, int blocks, float* us_per_launch) {
    MODEL(model);
    HIP_TRY(hipStreamSynchronize(M->stream));
    HIP_TRY(hipEventRecord(M->ev0, M->stream));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_noop, dim3(blocks), dim3(256), 0, M->stream, M->prob);
    HIP_TRY(hipEventRecord(M->ev1, M->stream));
    HIP_TRY(hipEventSynchronize(M->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, M->ev0, M->ev1));
    *us_per_launch = ms * 1000.f / n;
    return DNNCA_OK;
}

int dnnca_plan_dump(void* model, char* buf, size_t cap) {
    MODEL(model);
    if (!buf || !cap) return DNNCA_EINVAL;
    M->plan_text.clear();
    M->dry = true;
    dnnca_loss_cfg cfg = {0, 0.f, 0.f, 1.f};
    int B = M->desc.max_batch;
    M->defer_head = true;
    M->head_in_conv.requested = !getenv("DNNCA_NO_HEAD_IN_CONV");
    M->head_in_conv.y = M->y_stage;
    M->head_in_conv.cfg = cfg;
    int rc = M->forward(M->x_stage, B, true);
    M->defer_head = false;
    M->head_in_conv.requested = false;
    if (rc == DNNCA_OK) rc = M->loss_and_backward(M->y_stage, B, cfg, true);
    if (rc == DNNCA_OK) rc = M->optimizer_step(1e-3f);
    M->dry = false;
    snprintf(buf, cap, "%s", M->plan_text.c_str());
    return rc;
}

}  // extern "C"
