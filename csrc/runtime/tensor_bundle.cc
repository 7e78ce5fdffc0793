// TensorFlow-compatible checkpoint files ("TensorBundle", checkpoint format V2).
//
// What `checkpoint_dir=FLAGS.log_dir` at /root/reference/example.py:191 makes TensorFlow
// write: `<prefix>.index` + `<prefix>.data-0000i-of-0000N`.  This file writes and reads that
// format without TensorFlow:
//
//   *.data-0000i-of-0000N  raw little-endian tensor bytes, back to back, no framing
//   *.index                a LevelDB-style immutable sorted string table:
//        data blocks   entries { varint shared, varint non_shared, varint value_len, key suffix, value },
//                      a restart point every 16 keys, then u32 restart offsets and u32 count;
//                      each block is followed by 1 byte compression type (0) + u32 masked crc32c
//        metaindex     an empty block
//        index block   one entry per data block: separator key -> BlockHandle(varint offset, varint size)
//        footer        metaindex handle, index handle, zero padding to 40 bytes, magic 0xdb4775248b80fb57
//      key ""      -> BundleHeaderProto { num_shards = 1:N, endianness = 2 (LITTLE = 0, omitted),
//                                        version = 3 { producer = 1 } }
//      key <name>  -> BundleEntryProto  { dtype = 1, shape = 2 { dim = 2 { size = 1 } }, shard_id = 3,
//                                        offset = 4, size = 5, crc32c = 6 (fixed32, masked) }
//
// Sharded saves: every variable owner writes its own data shard plus a partial index
// (`<prefix>.index.part-<i>`); the chief merges the partial indices into `<prefix>.index`
// (what TF's MergeV2Checkpoints does).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

#include "crc32c.h"

namespace {

thread_local std::string g_error;

void set_error(const std::string& e) { g_error = e; }

// ---------------------------------------------------------------------------------- coding
void put_varint64(std::string* dst, uint64_t v) {
    while (v >= 128) {
        dst->push_back(static_cast<char>((v & 0x7f) | 0x80));
        v >>= 7;
    }
    dst->push_back(static_cast<char>(v));
}
void put_varint32(std::string* dst, uint32_t v) { put_varint64(dst, v); }
void put_fixed32(std::string* dst, uint32_t v) { dst->append(reinterpret_cast<const char*>(&v), 4); }
void put_fixed64(std::string* dst, uint64_t v) { dst->append(reinterpret_cast<const char*>(&v), 8); }

bool get_varint64(const uint8_t** p, const uint8_t* end, uint64_t* v) {
    uint64_t r = 0;
    for (int shift = 0; shift <= 63 && *p < end; shift += 7) {
        uint64_t b = *(*p)++;
        r |= (b & 0x7f) << shift;
        if (!(b & 0x80)) {
            *v = r;
            return true;
        }
    }
    return false;
}

// ---------------------------------------------------------------------------------- table builder
struct BlockBuilder {
    std::string buf;
    std::vector<uint32_t> restarts{0};
    int counter = 0;
    std::string last_key;
    static constexpr int kRestartInterval = 16;

    bool empty() const { return buf.empty(); }
    size_t size_estimate() const { return buf.size() + restarts.size() * 4 + 4; }

    void add(const std::string& key, const std::string& value) {
        size_t shared = 0;
        if (counter < kRestartInterval) {
            const size_t n = std::min(last_key.size(), key.size());
            while (shared < n && last_key[shared] == key[shared]) ++shared;
        } else {
            restarts.push_back(static_cast<uint32_t>(buf.size()));
            counter = 0;
        }
        put_varint32(&buf, static_cast<uint32_t>(shared));
        put_varint32(&buf, static_cast<uint32_t>(key.size() - shared));
        put_varint32(&buf, static_cast<uint32_t>(value.size()));
        buf.append(key.data() + shared, key.size() - shared);
        buf.append(value);
        last_key = key;
        ++counter;
    }

    std::string finish() {
        for (uint32_t r : restarts) put_fixed32(&buf, r);
        put_fixed32(&buf, static_cast<uint32_t>(restarts.size()));
        return buf;
    }

    void reset() {
        buf.clear();
        restarts.assign(1, 0);
        counter = 0;
        last_key.clear();
    }
};

struct BlockHandle {
    uint64_t offset = 0, size = 0;
    void encode(std::string* dst) const {
        put_varint64(dst, offset);
        put_varint64(dst, size);
    }
};

// bytewise-comparator key shortening, as LevelDB does for index keys
void find_shortest_separator(std::string* start, const std::string& limit) {
    const size_t min_len = std::min(start->size(), limit.size());
    size_t diff = 0;
    while (diff < min_len && (*start)[diff] == limit[diff]) ++diff;
    if (diff >= min_len) return;  // one is a prefix of the other
    const uint8_t b = static_cast<uint8_t>((*start)[diff]);
    if (b < 0xff && b + 1 < static_cast<uint8_t>(limit[diff])) {
        (*start)[diff] = static_cast<char>(b + 1);
        start->resize(diff + 1);
    }
}
void find_short_successor(std::string* key) {
    for (size_t i = 0; i < key->size(); ++i) {
        const uint8_t b = static_cast<uint8_t>((*key)[i]);
        if (b != 0xff) {
            (*key)[i] = static_cast<char>(b + 1);
            key->resize(i + 1);
            return;
        }
    }
}

struct TableBuilder {
    std::string out;          // whole file image (index files are small: one entry per variable)
    BlockBuilder data, index;
    std::string last_key;
    bool pending_index = false;
    BlockHandle pending_handle;
    static constexpr size_t kBlockSize = 262144;

    void write_block(const std::string& contents, BlockHandle* h) {
        h->offset = out.size();
        h->size = contents.size();
        out.append(contents);
        const char type = 0;  // no compression
        uint32_t crc = dtf::crc32c(contents.data(), contents.size());
        crc = dtf::crc32c_extend(crc, &type, 1);
        out.push_back(type);
        put_fixed32(&out, dtf::crc32c_mask(crc));
    }

    void flush() {
        if (data.empty()) return;
        write_block(data.finish(), &pending_handle);
        data.reset();
        pending_index = true;
    }

    void add(const std::string& key, const std::string& value) {
        if (pending_index) {
            find_shortest_separator(&last_key, key);
            std::string enc;
            pending_handle.encode(&enc);
            index.add(last_key, enc);
            pending_index = false;
        }
        last_key = key;
        data.add(key, value);
        if (data.size_estimate() >= kBlockSize) flush();
    }

    std::string finish() {
        flush();
        BlockHandle meta_h, index_h;
        BlockBuilder meta;
        write_block(meta.finish(), &meta_h);
        if (pending_index) {
            find_short_successor(&last_key);
            std::string enc;
            pending_handle.encode(&enc);
            index.add(last_key, enc);
            pending_index = false;
        }
        write_block(index.finish(), &index_h);
        std::string footer;
        meta_h.encode(&footer);
        index_h.encode(&footer);
        footer.resize(40, '\0');
        put_fixed64(&footer, 0xdb4775248b80fb57ull);
        out.append(footer);
        return out;
    }
};

// ---------------------------------------------------------------------------------- protos
struct Entry {
    int dtype = 0;
    std::vector<int64_t> dims;
    int shard_id = 0;
    int64_t offset = 0, size = 0;
    uint32_t crc = 0;  // masked
};

void tag(std::string* s, int field, int wire) { put_varint32(s, static_cast<uint32_t>((field << 3) | wire)); }

std::string encode_header(int num_shards) {
    std::string s;
    if (num_shards != 0) {
        tag(&s, 1, 0);
        put_varint64(&s, static_cast<uint64_t>(num_shards));
    }
    std::string ver;
    tag(&ver, 1, 0);
    put_varint64(&ver, 1);  // producer = kTensorBundleVersion
    tag(&s, 3, 2);
    put_varint32(&s, static_cast<uint32_t>(ver.size()));
    s.append(ver);
    return s;
}

std::string encode_entry(const Entry& e) {
    std::string s;
    if (e.dtype != 0) {
        tag(&s, 1, 0);
        put_varint64(&s, static_cast<uint64_t>(e.dtype));
    }
    std::string shape;
    for (int64_t d : e.dims) {
        std::string dim;
        if (d != 0) {
            tag(&dim, 1, 0);
            put_varint64(&dim, static_cast<uint64_t>(d));
        }
        tag(&shape, 2, 2);
        put_varint32(&shape, static_cast<uint32_t>(dim.size()));
        shape.append(dim);
    }
    tag(&s, 2, 2);
    put_varint32(&s, static_cast<uint32_t>(shape.size()));
    s.append(shape);
    if (e.shard_id != 0) {
        tag(&s, 3, 0);
        put_varint64(&s, static_cast<uint64_t>(e.shard_id));
    }
    if (e.offset != 0) {
        tag(&s, 4, 0);
        put_varint64(&s, static_cast<uint64_t>(e.offset));
    }
    if (e.size != 0) {
        tag(&s, 5, 0);
        put_varint64(&s, static_cast<uint64_t>(e.size));
    }
    if (e.crc != 0) {
        tag(&s, 6, 5);
        put_fixed32(&s, e.crc);
    }
    return s;
}

// minimal proto walker: calls f(field, wire, varint value or fixed value, payload ptr, payload len)
template <class F>
bool walk_proto(const uint8_t* p, const uint8_t* end, F f) {
    while (p < end) {
        uint64_t key;
        if (!get_varint64(&p, end, &key)) return false;
        const int field = static_cast<int>(key >> 3), wire = static_cast<int>(key & 7);
        if (wire == 0) {
            uint64_t v;
            if (!get_varint64(&p, end, &v)) return false;
            f(field, wire, v, nullptr, 0);
        } else if (wire == 2) {
            uint64_t n;
            if (!get_varint64(&p, end, &n) || static_cast<uint64_t>(end - p) < n) return false;
            f(field, wire, 0, p, static_cast<size_t>(n));
            p += n;
        } else if (wire == 5) {
            if (end - p < 4) return false;
            uint32_t v;
            std::memcpy(&v, p, 4);
            p += 4;
            f(field, wire, v, nullptr, 0);
        } else if (wire == 1) {
            if (end - p < 8) return false;
            uint64_t v;
            std::memcpy(&v, p, 8);
            p += 8;
            f(field, wire, v, nullptr, 0);
        } else {
            return false;
        }
    }
    return true;
}

bool decode_entry(const std::string& s, Entry* e) {
    const auto* p = reinterpret_cast<const uint8_t*>(s.data());
    return walk_proto(p, p + s.size(), [&](int field, int, uint64_t v, const uint8_t* pl, size_t n) {
        switch (field) {
            case 1: e->dtype = static_cast<int>(v); break;
            case 2:
                walk_proto(pl, pl + n, [&](int f2, int, uint64_t, const uint8_t* dp, size_t dn) {
                    if (f2 == 2) {
                        int64_t size = 0;
                        walk_proto(dp, dp + dn, [&](int f3, int, uint64_t v3, const uint8_t*, size_t) {
                            if (f3 == 1) size = static_cast<int64_t>(v3);
                        });
                        e->dims.push_back(size);
                    }
                });
                break;
            case 3: e->shard_id = static_cast<int>(v); break;
            case 4: e->offset = static_cast<int64_t>(v); break;
            case 5: e->size = static_cast<int64_t>(v); break;
            case 6: e->crc = static_cast<uint32_t>(v); break;
            default: break;
        }
    });
}

bool decode_header(const std::string& s, int* num_shards) {
    *num_shards = 0;
    const auto* p = reinterpret_cast<const uint8_t*>(s.data());
    return walk_proto(p, p + s.size(), [&](int field, int, uint64_t v, const uint8_t*, size_t) {
        if (field == 1) *num_shards = static_cast<int>(v);
    });
}

// ---------------------------------------------------------------------------------- files
std::string data_path(const std::string& prefix, int shard, int num_shards) {
    char buf[64];
    std::snprintf(buf, sizeof(buf), ".data-%05d-of-%05d", shard, num_shards);
    return prefix + buf;
}

bool write_file_atomic(const std::string& path, const std::string& contents) {
    const std::string tmp = path + ".tmp" + std::to_string(::getpid());
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) {
        set_error("cannot open " + tmp);
        return false;
    }
    bool ok = contents.empty() || std::fwrite(contents.data(), 1, contents.size(), f) == contents.size();
    ok = (std::fflush(f) == 0) && ok;
    ok = (std::fclose(f) == 0) && ok;
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) {
        set_error("cannot write " + path);
        std::remove(tmp.c_str());
        return false;
    }
    return true;
}

bool read_file(const std::string& path, std::string* out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) {
        set_error("cannot open " + path);
        return false;
    }
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out->assign(static_cast<size_t>(n > 0 ? n : 0), '\0');
    bool ok = n <= 0 || std::fread(&(*out)[0], 1, static_cast<size_t>(n), f) == static_cast<size_t>(n);
    std::fclose(f);
    if (!ok) set_error("short read on " + path);
    return ok;
}

// ---------------------------------------------------------------------------------- writer
struct Writer {
    std::string prefix;
    int num_shards = 1;
    int only_shard = -1;  // >= 0: partial writer owning exactly this shard
    std::map<int, FILE*> files;
    std::map<int, std::string> tmp_names;
    std::map<int, int64_t> offsets;
    std::map<std::string, Entry> entries;

    FILE* file_for(int shard) {
        auto it = files.find(shard);
        if (it != files.end()) return it->second;
        const std::string tmp = data_path(prefix, shard, num_shards) + ".tmp" + std::to_string(::getpid());
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f) {
            set_error("cannot open " + tmp);
            return nullptr;
        }
        files[shard] = f;
        tmp_names[shard] = tmp;
        offsets[shard] = 0;
        return f;
    }

    bool add(const std::string& name, int dtype, const int64_t* dims, int ndims, const void* data, int64_t nbytes,
             int shard) {
        if (name.empty()) {
            set_error("empty tensor name");
            return false;
        }
        if (entries.count(name)) {
            set_error("duplicate tensor " + name);
            return false;
        }
        if (shard < 0 || shard >= num_shards || (only_shard >= 0 && shard != only_shard)) {
            set_error("bad shard id for " + name);
            return false;
        }
        FILE* f = file_for(shard);
        if (!f) return false;
        if (nbytes > 0 && std::fwrite(data, 1, static_cast<size_t>(nbytes), f) != static_cast<size_t>(nbytes)) {
            set_error("short write for " + name);
            return false;
        }
        Entry e;
        e.dtype = dtype;
        e.dims.assign(dims, dims + ndims);
        e.shard_id = shard;
        e.offset = offsets[shard];
        e.size = nbytes;
        e.crc = dtf::crc32c_mask(dtf::crc32c(data, static_cast<size_t>(nbytes)));
        offsets[shard] += nbytes;
        entries[name] = e;
        return true;
    }

    bool close_data() {
        bool ok = true;
        for (auto& kv : files) {
            ok = (std::fflush(kv.second) == 0) && ok;
            ok = (std::fclose(kv.second) == 0) && ok;
            const std::string final_name = data_path(prefix, kv.first, num_shards);
            ok = (std::rename(tmp_names[kv.first].c_str(), final_name.c_str()) == 0) && ok;
        }
        files.clear();
        if (!ok) set_error("cannot finalise data shards of " + prefix);
        return ok;
    }

    bool finish() {
        // a shard that received no tensor still gets its (empty) data file, as TF writes one per shard
        for (int s = 0; s < num_shards; ++s)
            if (only_shard < 0 || s == only_shard)
                if (!file_for(s)) return false;
        if (!close_data()) return false;
        if (only_shard >= 0) {
            // partial index: plain list of (name, entry) records for the chief's merge
            std::string out;
            for (auto& kv : entries) {
                const std::string enc = encode_entry(kv.second);
                put_varint32(&out, static_cast<uint32_t>(kv.first.size()));
                out.append(kv.first);
                put_varint32(&out, static_cast<uint32_t>(enc.size()));
                out.append(enc);
            }
            return write_file_atomic(prefix + ".index.part-" + std::to_string(only_shard), out);
        }
        TableBuilder tb;
        tb.add("", encode_header(num_shards));
        for (auto& kv : entries) tb.add(kv.first, encode_entry(kv.second));
        return write_file_atomic(prefix + ".index", tb.finish());
    }
};

bool merge_partials(const std::string& prefix, int num_shards) {
    std::map<std::string, std::string> all;
    for (int s = 0; s < num_shards; ++s) {
        const std::string part = prefix + ".index.part-" + std::to_string(s);
        std::string buf;
        if (!read_file(part, &buf)) return false;
        const auto* p = reinterpret_cast<const uint8_t*>(buf.data());
        const auto* end = p + buf.size();
        while (p < end) {
            uint64_t kn, vn;
            if (!get_varint64(&p, end, &kn) || static_cast<uint64_t>(end - p) < kn) { set_error("corrupt " + part); return false; }
            std::string k(reinterpret_cast<const char*>(p), static_cast<size_t>(kn));
            p += kn;
            if (!get_varint64(&p, end, &vn) || static_cast<uint64_t>(end - p) < vn) { set_error("corrupt " + part); return false; }
            std::string v(reinterpret_cast<const char*>(p), static_cast<size_t>(vn));
            p += vn;
            if (all.count(k)) { set_error("tensor " + k + " saved by two shards"); return false; }
            all[k] = v;
        }
    }
    TableBuilder tb;
    tb.add("", encode_header(num_shards));
    for (auto& kv : all) tb.add(kv.first, kv.second);
    if (!write_file_atomic(prefix + ".index", tb.finish())) return false;
    for (int s = 0; s < num_shards; ++s) std::remove((prefix + ".index.part-" + std::to_string(s)).c_str());
    return true;
}

// ---------------------------------------------------------------------------------- reader
struct Reader {
    std::string prefix;
    int num_shards = 0;
    std::vector<std::pair<std::string, Entry>> entries;

    static bool check_block(const std::string& file, uint64_t off, uint64_t size, std::string* contents) {
        if (off + size + 5 > file.size()) {
            set_error("block handle out of range");
            return false;
        }
        const char* p = file.data() + off;
        uint32_t stored;
        std::memcpy(&stored, p + size + 1, 4);
        const uint32_t actual = dtf::crc32c_mask(dtf::crc32c(p, static_cast<size_t>(size) + 1));
        if (stored != actual) {
            set_error("block checksum mismatch in index");
            return false;
        }
        if (p[size] != 0) {
            set_error("compressed index blocks are not supported");
            return false;
        }
        contents->assign(p, static_cast<size_t>(size));
        return true;
    }

    template <class F>
    static bool iterate_block(const std::string& block, F f) {
        if (block.size() < 4) return false;
        uint32_t nrestarts;
        std::memcpy(&nrestarts, block.data() + block.size() - 4, 4);
        const size_t data_end = block.size() - 4 - static_cast<size_t>(nrestarts) * 4;
        if (data_end > block.size()) return false;
        const auto* p = reinterpret_cast<const uint8_t*>(block.data());
        const auto* end = p + data_end;
        std::string key;
        while (p < end) {
            uint64_t shared, non_shared, vlen;
            if (!get_varint64(&p, end, &shared) || !get_varint64(&p, end, &non_shared) || !get_varint64(&p, end, &vlen))
                return false;
            if (shared > key.size() || static_cast<uint64_t>(end - p) < non_shared + vlen) return false;
            key.resize(static_cast<size_t>(shared));
            key.append(reinterpret_cast<const char*>(p), static_cast<size_t>(non_shared));
            p += non_shared;
            f(key, std::string(reinterpret_cast<const char*>(p), static_cast<size_t>(vlen)));
            p += vlen;
        }
        return true;
    }

    bool open(const std::string& pfx) {
        prefix = pfx;
        std::string file;
        if (!read_file(prefix + ".index", &file)) return false;
        if (file.size() < 48) {
            set_error("index file too short");
            return false;
        }
        const char* footer = file.data() + file.size() - 48;
        uint64_t magic;
        std::memcpy(&magic, footer + 40, 8);
        if (magic != 0xdb4775248b80fb57ull) {
            set_error("bad table magic");
            return false;
        }
        const auto* p = reinterpret_cast<const uint8_t*>(footer);
        const auto* end = p + 40;
        uint64_t mo, ms, io, is;
        if (!get_varint64(&p, end, &mo) || !get_varint64(&p, end, &ms) || !get_varint64(&p, end, &io) ||
            !get_varint64(&p, end, &is)) {
            set_error("bad footer");
            return false;
        }
        std::string index_block;
        if (!check_block(file, io, is, &index_block)) return false;
        std::vector<std::pair<uint64_t, uint64_t>> handles;
        bool ok = iterate_block(index_block, [&](const std::string&, const std::string& v) {
            const auto* q = reinterpret_cast<const uint8_t*>(v.data());
            const auto* qe = q + v.size();
            uint64_t o = 0, s = 0;
            get_varint64(&q, qe, &o);
            get_varint64(&q, qe, &s);
            handles.emplace_back(o, s);
        });
        if (!ok) {
            set_error("corrupt index block");
            return false;
        }
        bool saw_header = false;
        for (auto& h : handles) {
            std::string block;
            if (!check_block(file, h.first, h.second, &block)) return false;
            bool good = true;
            ok = iterate_block(block, [&](const std::string& k, const std::string& v) {
                if (k.empty()) {
                    saw_header = decode_header(v, &num_shards);
                } else {
                    Entry e;
                    if (!decode_entry(v, &e)) good = false;
                    entries.emplace_back(k, e);
                }
            });
            if (!ok || !good) {
                set_error("corrupt data block");
                return false;
            }
        }
        if (!saw_header) {
            set_error("missing bundle header");
            return false;
        }
        return true;
    }

    const Entry* find(const std::string& name) const {
        for (auto& kv : entries)
            if (kv.first == name) return &kv.second;
        return nullptr;
    }

    bool read(const std::string& name, void* buf, int64_t cap, bool verify) {
        const Entry* e = find(name);
        if (!e) {
            set_error("tensor not found: " + name);
            return false;
        }
        if (e->size > cap) {
            set_error("buffer too small for " + name);
            return false;
        }
        const std::string path = data_path(prefix, e->shard_id, num_shards);
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) {
            set_error("cannot open " + path);
            return false;
        }
        bool ok = std::fseek(f, static_cast<long>(e->offset), SEEK_SET) == 0 &&
                  (e->size == 0 || std::fread(buf, 1, static_cast<size_t>(e->size), f) == static_cast<size_t>(e->size));
        std::fclose(f);
        if (!ok) {
            set_error("short read for " + name);
            return false;
        }
        if (verify && dtf::crc32c_mask(dtf::crc32c(buf, static_cast<size_t>(e->size))) != e->crc) {
            set_error("crc32c mismatch for tensor " + name);
            return false;
        }
        return true;
    }
};

}  // namespace

DTF_EXPORT const char* dtf_last_error() { return g_error.c_str(); }

DTF_EXPORT void* dtf_bundle_writer_create(const char* prefix, int num_shards, int only_shard) {
    auto* w = new Writer();
    w->prefix = prefix;
    w->num_shards = num_shards < 1 ? 1 : num_shards;
    w->only_shard = only_shard;
    return w;
}
DTF_EXPORT int dtf_bundle_writer_add(void* h, const char* name, int dtype, const int64_t* dims, int ndims,
                                     const void* data, int64_t nbytes, int shard) {
    return static_cast<Writer*>(h)->add(name, dtype, dims, ndims, data, nbytes, shard) ? 0 : -1;
}
DTF_EXPORT int dtf_bundle_writer_finish(void* h) {
    auto* w = static_cast<Writer*>(h);
    const bool ok = w->finish();
    delete w;
    return ok ? 0 : -1;
}
DTF_EXPORT void dtf_bundle_writer_abort(void* h) {
    auto* w = static_cast<Writer*>(h);
    for (auto& kv : w->files) {
        std::fclose(kv.second);
        std::remove(w->tmp_names[kv.first].c_str());
    }
    delete w;
}
DTF_EXPORT int dtf_bundle_merge(const char* prefix, int num_shards) { return merge_partials(prefix, num_shards) ? 0 : -1; }

DTF_EXPORT void* dtf_bundle_reader_open(const char* prefix) {
    auto r = std::make_unique<Reader>();
    if (!r->open(prefix)) return nullptr;
    return r.release();
}
DTF_EXPORT void dtf_bundle_reader_close(void* h) { delete static_cast<Reader*>(h); }
DTF_EXPORT int dtf_bundle_reader_num_entries(void* h) { return static_cast<int>(static_cast<Reader*>(h)->entries.size()); }
DTF_EXPORT int dtf_bundle_reader_num_shards(void* h) { return static_cast<Reader*>(h)->num_shards; }
DTF_EXPORT int dtf_bundle_reader_entry(void* h, int i, char* name, int name_cap, int* dtype, int64_t* dims, int dims_cap,
                                       int* ndims, int64_t* nbytes, int* shard, int64_t* offset, uint32_t* crc) {
    auto* r = static_cast<Reader*>(h);
    if (i < 0 || i >= static_cast<int>(r->entries.size())) return -1;
    const auto& kv = r->entries[static_cast<size_t>(i)];
    if (static_cast<int>(kv.first.size()) + 1 > name_cap || static_cast<int>(kv.second.dims.size()) > dims_cap) return -1;
    std::memcpy(name, kv.first.c_str(), kv.first.size() + 1);
    *dtype = kv.second.dtype;
    *ndims = static_cast<int>(kv.second.dims.size());
    for (size_t d = 0; d < kv.second.dims.size(); ++d) dims[d] = kv.second.dims[d];
    *nbytes = kv.second.size;
    *shard = kv.second.shard_id;
    if (offset) *offset = kv.second.offset;
    if (crc) *crc = kv.second.crc;
    return 0;
}
DTF_EXPORT int dtf_bundle_reader_read(void* h, const char* name, void* buf, int64_t cap, int verify_crc) {
    return static_cast<Reader*>(h)->read(name, buf, cap, verify_crc != 0) ? 0 : -1;
}
