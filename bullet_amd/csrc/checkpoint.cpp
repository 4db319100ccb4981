// Checkpoints and saved network files, host side (section 6 of include/bullet_amd.h).
// Formats restated from crates/trainer/src/model/weights.rs:95-177,183-334,404-467,
// crates/trainer/src/optimiser.rs:218-236, optimiser/adam.rs:300-324 and crates/bullet_lib/src/value/save.rs.
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "common.h"

namespace {

int dense_in_of(const bt_net_desc *d, int k) {
    if (k == 0) return d->perspectives * (d->pairwise ? d->ft_out / 2 : d->ft_out);
    return (d->dense_out[k - 1] - d->dense_skip[k - 1]) * (d->dense_dual[k - 1] ? 2 : 1);
}

int find_tensor(const bt_net_desc *d, const std::string &id) {
    for (int t = 0; t < bt_net_num_tensors(d); t++)
        if (id == bt_net_tensor_name(d, t)) return t;
    return -1;
}

int write_file(const char *path, const std::vector<unsigned char> &buf) {
    FILE *f = std::fopen(path, "wb");
    BT_REQUIRE(f, "cannot create %s", path);
    const size_t w = buf.empty() ? 0 : std::fwrite(buf.data(), 1, buf.size(), f);
    const int rc = std::fclose(f);
    BT_REQUIRE(w == buf.size() && rc == 0, "short write to %s", path);
    return BT_OK;
}

int read_file(const char *path, std::vector<unsigned char> *buf) {
    FILE *f = std::fopen(path, "rb");
    BT_REQUIRE(f, "cannot open %s", path);
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    buf->resize(n > 0 ? (size_t)n : 0);
    const size_t r = buf->empty() ? 0 : std::fread(buf->data(), 1, buf->size(), f);
    std::fclose(f);
    BT_REQUIRE(r == buf->size(), "short read from %s", path);
    return BT_OK;
}

// utils::write_to_byte_buffer, weights.rs:404-431
void append_record(std::vector<unsigned char> *buf, const char *id, const float *v, size_t n) {
    for (const char *c = id; *c; c++) buf->push_back((unsigned char)*c);
    buf->push_back('\n');
    const uint64_t n64 = n;
    for (int i = 0; i < 8; i++) buf->push_back((unsigned char)(n64 >> (8 * i)));
    const unsigned char *p = reinterpret_cast<const unsigned char *>(v);  // little-endian host (x86-64)
    buf->insert(buf->end(), p, p + n * sizeof(float));
}

// QuantTarget::quantise, weights.rs:288-331: q * x in f64, round (half away from zero) or truncate, then the
// value must survive the cast to the target type
template <typename T>
int quantise_to(const std::vector<float> &v, int q, bool round, const char *tname, std::vector<unsigned char> *out) {
    for (float x : v) {
        const double qf0 = (double)q * (double)x;
        const double qf = round ? std::round(qf0) : std::trunc(qf0);
        // Rust `as` saturates and maps NaN to 0; the comparison then rejects anything that did not fit
        T y;
        if (std::isnan(qf)) y = 0;
        else if (qf <= (double)std::numeric_limits<T>::min()) y = std::numeric_limits<T>::min();
        else if (qf >= (double)std::numeric_limits<T>::max()) y = std::numeric_limits<T>::max();
        else y = (T)qf;
        BT_REQUIRE(qf == (double)y, "Failed quantisation from f32 to %s!", tname);
        const unsigned char *p = reinterpret_cast<const unsigned char *>(&y);
        out->insert(out->end(), p, p + sizeof(T));
    }
    return BT_OK;
}

int entry_bytes(const bt_net_desc *d, const float *flat, const bt_saved_format &f, std::vector<unsigned char> *out) {
    if (f.custom || !f.id) {  // SavedFormat::custom
        const unsigned char *p = static_cast<const unsigned char *>(f.custom);
        if (p) out->insert(out->end(), p, p + f.custom_len);
        return BT_OK;
    }
    const int t = find_tensor(d, f.id);
    BT_REQUIRE(t >= 0, "No weight \"%s\"!", f.id);
    const size_t n = bt_net_tensor_size(d, t);
    std::vector<float> v(flat + bt_net_tensor_offset(d, t), flat + bt_net_tensor_offset(d, t) + n);
    if (f.merge_factoriser) {
        BT_REQUIRE(d->ft_factoriser > 0 && t == 0, "merge_factoriser applies to l0w of a net with a factoriser");
        const int tf = bt_net_num_tensors(d) - 1;
        const float *wf = flat + bt_net_tensor_offset(d, tf);
        const size_t nf = bt_net_tensor_size(d, tf);
        for (size_t i = 0; i < n; i++) v[i] = v[i] + wf[i % nf];  // weights + factoriser.repeat(NB)
    }
    if (f.transform)
        BT_REQUIRE(f.transform(f.user, f.id, v.data(), v.size()) == 0, "save transform of \"%s\" failed", f.id);
    if (f.transpose) {  // SavedFormat::transpose_impl, weights.rs:262-275
        int rows = 0, cols = 0;
        bt_net_tensor_shape(d, t, &rows, &cols);
        std::vector<float> w(v.size());
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols; j++) w[(size_t)cols * i + j] = v[(size_t)rows * j + i];
        v.swap(w);
    }
    switch (f.quant) {
    case BT_QUANT_FLOAT: {
        const unsigned char *p = reinterpret_cast<const unsigned char *>(v.data());
        out->insert(out->end(), p, p + v.size() * sizeof(float));
        return BT_OK;
    }
    case BT_QUANT_I8:
        BT_REQUIRE(f.multiplier >= -32768 && f.multiplier <= 32767, "i8 multiplier must fit i16");
        return quantise_to<int8_t>(v, f.multiplier, f.round != 0, "i8", out);
    case BT_QUANT_I16:
        BT_REQUIRE(f.multiplier >= -32768 && f.multiplier <= 32767, "i16 multiplier must fit i16");
        return quantise_to<int16_t>(v, f.multiplier, f.round != 0, "i16", out);
    case BT_QUANT_I32: return quantise_to<int32_t>(v, f.multiplier, f.round != 0, "i32", out);
    default: BT_REQUIRE(false, "unknown quantisation target %d", f.quant);
    }
    return BT_OK;
}

}  // namespace

extern "C" {

const char *bt_net_tensor_name(const bt_net_desc *d, int t) {
    static const char *names[] = {"l0w", "l0b", "l1w", "l1b", "l2w", "l2b", "l3w", "l3b", "l4w", "l4b"};
    if (!d || t < 0 || t >= bt_net_num_tensors(d)) return "";
    if (t == 2 + 2 * d->n_dense) return "l0f";
    return t < 10 ? names[t] : "";
}

int bt_net_tensor_shape(const bt_net_desc *d, int t, int *rows, int *cols) {
    BT_REQUIRE(d && rows && cols && t >= 0 && t < bt_net_num_tensors(d), "bt_net_tensor_shape: bad arguments");
    if (t == 2 + 2 * d->n_dense) {
        *rows = d->ft_out, *cols = d->num_inputs / d->ft_factoriser;
    } else if (t == 0) {
        *rows = d->ft_out, *cols = d->num_inputs;
    } else if (t == 1) {
        *rows = d->ft_out, *cols = 1;
    } else {
        const int k = (t - 2) / 2;
        *rows = d->dense_out[k] * (d->dense_bucketed[k] ? d->num_buckets : 1);
        *cols = (t & 1) ? 1 : dense_in_of(d, k);
    }
    return BT_OK;
}

int bt_checkpoint_write(const bt_net_desc *d, const float *flat, const char *path) {
    BT_REQUIRE(d && flat && path, "bt_checkpoint_write: null argument");
    std::vector<int> order(bt_net_num_tensors(d));
    for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
    std::sort(order.begin(), order.end(), [&](int a, int b) {  // BTreeMap<String, _> iteration order
        return std::strcmp(bt_net_tensor_name(d, a), bt_net_tensor_name(d, b)) < 0;
    });
    std::vector<unsigned char> buf;
    for (int t : order) append_record(&buf, bt_net_tensor_name(d, t), flat + bt_net_tensor_offset(d, t), bt_net_tensor_size(d, t));
    return write_file(path, buf);
}

int bt_checkpoint_read(const bt_net_desc *d, float *flat, const char *path) {
    BT_REQUIRE(d && flat && path, "bt_checkpoint_read: null argument");
    std::vector<unsigned char> buf;
    if (int rc = read_file(path, &buf)) return rc;
    size_t off = 0;
    while (off < buf.size()) {  // utils::read_from_byte_buffer, weights.rs:433-467
        std::string id;
        while (off < buf.size() && buf[off] != '\n') id.push_back((char)buf[off++]);
        BT_REQUIRE(off < buf.size(), "%s: truncated record id", path);
        off++;
        BT_REQUIRE(off + 8 <= buf.size(), "%s: truncated record header", path);
        uint64_t n = 0;
        for (int i = 0; i < 8; i++) n |= (uint64_t)buf[off + i] << (8 * i);
        off += 8;
        const int t = find_tensor(d, id);
        BT_REQUIRE(t >= 0, "No weight with ID found! (\"%s\")", id.c_str());
        BT_REQUIRE(n == bt_net_tensor_size(d, t), "Invalid buffer size! (\"%s\": %llu, expected %zu)", id.c_str(),
                   (unsigned long long)n, bt_net_tensor_size(d, t));
        BT_REQUIRE(off + n * 4 <= buf.size(), "%s: truncated record \"%s\"", path, id.c_str());
        std::memcpy(flat + bt_net_tensor_offset(d, t), buf.data() + off, n * 4);
        off += n * 4;
    }
    return BT_OK;
}

int bt_save_quantised(const bt_net_desc *d, const float *flat, const bt_saved_format *fmt, int n_fmt, const char *path) {
    BT_REQUIRE(d && flat && path && (fmt || n_fmt == 0), "bt_save_quantised: null argument");
    std::vector<unsigned char> buf;
    for (int i = 0; i < n_fmt; i++)
        if (int rc = entry_bytes(d, flat, fmt[i], &buf)) return rc;
    const size_t rem = buf.size() % 64;  // ModelWeights::to_quantised_buffer(pad = true), weights.rs:160-177
    if (rem > 0) {
        static const char chs[] = {'b', 'u', 'l', 'l', 'e', 't'};
        for (size_t i = 0; i < 64 - rem; i++) buf.push_back((unsigned char)chs[i % 6]);
    }
    return write_file(path, buf);
}

int bt_save_raw(const bt_net_desc *d, const float *flat, const bt_saved_format *fmt, int n_fmt, const char *path) {
    BT_REQUIRE(d && flat && path && (fmt || n_fmt == 0), "bt_save_raw: null argument");
    std::vector<unsigned char> buf;
    for (int i = 0; i < n_fmt; i++) {
        if (!fmt[i].id || fmt[i].custom) continue;  // save_unquantised skips entries without an id, save.rs:58-64
        const int t = find_tensor(d, fmt[i].id);
        BT_REQUIRE(t >= 0, "No weight \"%s\"!", fmt[i].id);
        const unsigned char *p = reinterpret_cast<const unsigned char *>(flat + bt_net_tensor_offset(d, t));
        buf.insert(buf.end(), p, p + bt_net_tensor_size(d, t) * sizeof(float));
    }
    return write_file(path, buf);
}

}  // extern "C"
