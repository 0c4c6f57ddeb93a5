// Model directory -> Model, natively: the file side of ModelLoader::load_tts_model (qwen_tts/src/model/loader.rs:267-324).
// Reads config.json, speech_tokenizer/config.json and the safetensors files through read-only mappings and hands every tensor
// to Model::load_tensor under its HF key.  Like the reference, a sharded checkpoint contributes its first shard only
// (model/loader.rs:219-235), and encoder.* tensors are loaded only on request (LoaderConfig / with_encoder).
//
// safetensors layout: u64 little-endian header length n, n bytes of JSON {"key": {"dtype", "shape", "data_offsets": [b, e]}, ..,
// "__metadata__": {..}}, then the tensor bytes; offsets are relative to the first byte after the header.
#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <fstream>
#include <sstream>

#include "json_min.hpp"
#include "loader.hpp"

#include <cmath>

namespace qtts {

namespace {

struct Mapping {  // read-only mapping of a whole file
    const uint8_t* p = nullptr;
    size_t n = 0;
    explicit Mapping(const std::string& path) {
        const int fd = open(path.c_str(), O_RDONLY);
        QTTS_CHECK(fd >= 0, ERR_IO, "cannot open " + path);
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size < 8) {
            close(fd);
            throw Error(ERR_IO, "not a safetensors file: " + path);
        }
        n = (size_t)st.st_size;
        void* q = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        QTTS_CHECK(q != MAP_FAILED, ERR_IO, "cannot map " + path);
        p = (const uint8_t*)q;
    }
    ~Mapping() {
        if (p) munmap((void*)p, n);
    }
    Mapping(const Mapping&) = delete;
    Mapping& operator=(const Mapping&) = delete;
};

bool file_exists(const std::string& path) {
    struct stat st;
    return stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

std::string read_text(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    QTTS_CHECK(f.good(), ERR_IO, "cannot read " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {  // subnormal: normalise
            int e = -1;
            do {
                ++e;
                man <<= 1;
            } while ((man & 0x400u) == 0);
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

// every tensor of one safetensors file; returns how many were handed over
int load_file(Model& m, const std::string& path, bool load_encoder) {
    Mapping map(path);
    uint64_t hlen = 0;
    std::memcpy(&hlen, map.p, 8);
    QTTS_CHECK(hlen >= 2 && hlen <= map.n - 8, ERR_IO, "bad safetensors header in " + path);
    const Json header = JsonParser::parse(std::string((const char*)map.p + 8, (size_t)hlen));
    QTTS_CHECK(header.kind == Json::Obj, ERR_IO, "bad safetensors header in " + path);
    const uint8_t* data = map.p + 8 + hlen;
    const size_t data_n = map.n - 8 - (size_t)hlen;
    int count = 0;
    std::vector<float> widened;
    for (const auto& kvp : header.obj) {
        const std::string& key = kvp.first;
        if (key == "__metadata__") continue;
        if (!load_encoder && key.rfind("encoder.", 0) == 0) continue;
        const Json& t = kvp.second;
        const std::string dt = t.str_or("dtype", "");
        const Json& sh = t.at("shape");
        const Json& off = t.at("data_offsets");
        QTTS_CHECK(sh.kind == Json::Arr && off.kind == Json::Arr && off.arr.size() == 2, ERR_IO, "bad safetensors entry " + key);
        // header numbers come from the file: only exact non-negative integers below 2^53 are taken, products are checked
        auto as_count = [&](const Json& v, int64_t& out) {
            if (v.kind != Json::Num || !(v.num >= 0.0) || v.num > 9007199254740992.0 || v.num != std::floor(v.num)) return false;
            out = (int64_t)v.num;
            return true;
        };
        std::vector<int64_t> shape;
        int64_t n = 1;
        bool ok = true;
        for (const Json& d : sh.arr) {
            int64_t dim = 0;
            ok = ok && as_count(d, dim) && !__builtin_mul_overflow(n, dim, &n);
            shape.push_back(dim);
        }
        int64_t b64 = 0, e64 = 0, nbytes = 0;
        ok = ok && as_count(off.arr[0], b64) && as_count(off.arr[1], e64);
        const size_t es = dt == "F32" ? 4 : ((dt == "BF16" || dt == "F16") ? 2 : 0);
        if (es == 0) continue;  // integer / bool bookkeeping tensors: nothing on the path reads one
        ok = ok && !__builtin_mul_overflow(n, (int64_t)es, &nbytes);
        QTTS_CHECK(ok, ERR_IO, "bad shape or data offsets for tensor " + key + " in " + path);
        const size_t b = (size_t)b64, e = (size_t)e64;
        QTTS_CHECK(b <= e && e <= data_n && e - b == (size_t)nbytes, ERR_IO, "tensor " + key + " does not match its data range");
        if (shape.empty()) shape.push_back(1);  // a scalar is a vector of one
        if (dt == "F16") {
            widened.resize((size_t)n);
            for (int64_t i = 0; i < n; ++i) {
                uint16_t h;  // the data section starts wherever the header ends: no alignment to rely on
                std::memcpy(&h, data + b + (size_t)i * 2, 2);
                widened[(size_t)i] = f16_to_f32(h);
            }
            m.load_tensor(key, widened.data(), F32, shape.data(), (int)shape.size(), false);
        } else {
            m.load_tensor(key, data + b, dt == "F32" ? F32 : BF16, shape.data(), (int)shape.size(), false);
        }
        ++count;
    }
    return count;
}

}  // namespace

void load_model_dir(Model& m, const std::string& dir, bool load_tokenizer, bool load_encoder) {
    const std::string cfg_path = dir + "/config.json";
    QTTS_CHECK(file_exists(cfg_path), ERR_IO, "config.json not found in " + dir);
    const std::string codec_cfg_path = dir + "/speech_tokenizer/config.json";
    const bool codec = load_tokenizer && file_exists(codec_cfg_path);
    const std::string cfg = read_text(cfg_path);
    const std::string codec_cfg = codec ? read_text(codec_cfg_path) : std::string();
    m.parse_config(cfg, codec ? codec_cfg.c_str() : nullptr);
    std::string main = dir + "/model.safetensors";
    if (!file_exists(main)) {
        std::vector<std::string> shards;
        if (DIR* d = opendir(dir.c_str())) {
            while (dirent* ent = readdir(d)) {
                const std::string f = ent->d_name;
                if (f.rfind("model-", 0) == 0 && f.size() > 12 && f.compare(f.size() - 12, 12, ".safetensors") == 0) shards.push_back(f);
            }
            closedir(d);
        }
        QTTS_CHECK(!shards.empty(), ERR_IO, "no model.safetensors in " + dir);
        std::sort(shards.begin(), shards.end());
        main = dir + "/" + shards[0];
    }
    load_file(m, main, load_encoder);
    if (codec) {  // find_tokenizer_weights_file (model/loader.rs:241-257): the first of these that exists
        for (const char* cand : {"speech_tokenizer/model.safetensors", "tokenizer.safetensors", "audio_tokenizer.safetensors", "tokenizer/model.safetensors"}) {
            const std::string tok = dir + "/" + cand;
            if (file_exists(tok)) {
                load_file(m, tok, load_encoder);
                break;
            }
        }
    }
    m.finalize();
}

}  // namespace qtts
