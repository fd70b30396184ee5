// ark/Npz.h — tiny .npz (zip of .npy) reader for the SMPL model file the reference loads with cnpy
// (AvatarModel.cpp:23-127; cnpy.cpp).  Supports stored and deflate entries (zlib), little-endian f4/f8/i4/i8/u4/u8,
// C or Fortran order; values are converted to double / int64 in C order.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace ark {
namespace npz {

struct Array {
    std::vector<size_t> shape;
    std::vector<double> f;      // floating data (row-major)
    std::vector<long long> i;   // integer data (row-major)
    bool is_int = false;
    size_t size() const { size_t n = 1; for (size_t s : shape) n *= s; return n; }
    double at(size_t k) const { return is_int ? (double)i[k] : f[k]; }
};

inline Array parse_npy(const std::vector<unsigned char>& b) {
    if (b.size() < 10 || std::memcmp(b.data(), "\x93NUMPY", 6) != 0) throw std::runtime_error("npz: bad .npy magic");
    const int major = b[6];
    size_t hlen, off;
    if (major == 1) { hlen = b[8] | (b[9] << 8); off = 10; }
    else { if (b.size() < 12) throw std::runtime_error("npz: truncated .npy header"); hlen = b[8] | (b[9] << 8) | (b[10] << 16) | ((size_t)b[11] << 24); off = 12; }
    if (hlen > b.size() || off + hlen > b.size()) throw std::runtime_error("npz: .npy header longer than the member");
    const std::string hdr((const char*)b.data() + off, hlen);
    const size_t npos = std::string::npos;
    auto find = [&](const std::string& key) {
        const size_t p = hdr.find("'" + key + "'");
        if (p == npos) throw std::runtime_error("npz: header key missing: " + key);
        const size_t c = hdr.find(':', p);
        if (c == npos) throw std::runtime_error("npz: malformed header near " + key);
        return c + 1;
    };
    size_t p = find("descr");
    const size_t q0 = hdr.find('\'', p), q1 = q0 == npos ? npos : hdr.find('\'', q0 + 1);
    if (q0 == npos || q1 == npos) throw std::runtime_error("npz: unsupported dtype description (structured or object arrays)");
    const std::string descr = hdr.substr(q0 + 1, q1 - q0 - 1);
    p = find("fortran_order");
    const size_t fo = hdr.find_first_not_of(' ', p);
    if (fo == npos) throw std::runtime_error("npz: malformed fortran_order");
    const bool fortran = hdr.compare(fo, 4, "True") == 0;
    p = find("shape");
    const size_t s0 = hdr.find('(', p), s1 = s0 == npos ? npos : hdr.find(')', s0);
    if (s0 == npos || s1 == npos) throw std::runtime_error("npz: malformed shape");
    Array a;
    {
        std::string sh = hdr.substr(s0 + 1, s1 - s0 - 1);
        size_t pos = 0;
        while (pos < sh.size()) {
            while (pos < sh.size() && (sh[pos] == ' ' || sh[pos] == ',')) ++pos;
            if (pos >= sh.size()) break;
            if (sh[pos] < '0' || sh[pos] > '9') throw std::runtime_error("npz: malformed shape");
            a.shape.push_back(std::stoull(sh.substr(pos)));
            while (pos < sh.size() && sh[pos] != ',') ++pos;
        }
    }
    if (descr.size() < 3 || (descr[0] != '<' && descr[0] != '|' && descr[0] != '=')) throw std::runtime_error("npz: unsupported dtype " + descr);
    const char kind = descr[1];
    if (descr[2] < '0' || descr[2] > '9') throw std::runtime_error("npz: unsupported dtype " + descr);
    const int width = std::stoi(descr.substr(2));
    if (!((kind == 'f' && (width == 4 || width == 8)) || ((kind == 'i' || kind == 'u') && (width == 4 || width == 8))))
        throw std::runtime_error("npz: unsupported dtype " + descr);
    size_t n = 1;
    for (size_t sdim : a.shape) {      // the product must not wrap size_t (a crafted header with two 2^40 dimensions did)
        if (sdim > ((size_t)1 << 40) || (sdim != 0 && n > ((size_t)1 << 40) / sdim)) throw std::runtime_error("npz: implausible shape");
        n *= sdim;
    }
    const unsigned char* d = b.data() + off + hlen;
    if (n > (b.size() - off - hlen) / (size_t)width) throw std::runtime_error("npz: truncated array");
    a.is_int = (kind == 'i' || kind == 'u');
    std::vector<size_t> strideC(a.shape.size(), 1), strideF(a.shape.size(), 1);
    for (int k = (int)a.shape.size() - 2; k >= 0; --k) strideC[k] = strideC[k + 1] * a.shape[k + 1];
    for (size_t k = 1; k < a.shape.size(); ++k) strideF[k] = strideF[k - 1] * a.shape[k - 1];
    auto src_index = [&](size_t lin) {
        if (!fortran) return lin;
        size_t idx = 0;
        for (size_t k = 0; k < a.shape.size(); ++k) { const size_t c = (lin / strideC[k]) % a.shape[k]; idx += c * strideF[k]; }
        return idx;
    };
    if (a.is_int) a.i.resize(n); else a.f.resize(n);
    for (size_t lin = 0; lin < n; ++lin) {
        const unsigned char* e = d + src_index(lin) * width;
        if (kind == 'f' && width == 8) { double v; std::memcpy(&v, e, 8); a.f[lin] = v; }
        else if (kind == 'f' && width == 4) { float v; std::memcpy(&v, e, 4); a.f[lin] = v; }
        else if (kind == 'i' && width == 8) { int64_t v; std::memcpy(&v, e, 8); a.i[lin] = v; }
        else if (kind == 'i' && width == 4) { int32_t v; std::memcpy(&v, e, 4); a.i[lin] = v; }
        else if (kind == 'u' && width == 8) { uint64_t v; std::memcpy(&v, e, 8); a.i[lin] = (long long)v; }
        else if (kind == 'u' && width == 4) { uint32_t v; std::memcpy(&v, e, 4); a.i[lin] = v; }
        else throw std::runtime_error("npz: unsupported dtype " + descr);
    }
    return a;
}

/** Reads the members of `path`.  Every offset and length of the zip structures is checked against the file size (a
 *  truncated or corrupt file throws instead of reading out of bounds).  Members that are not numeric arrays (strings,
 *  objects, bools - SMPL exports carry some) are SKIPPED, as cnpy tolerates them; a caller fails only when a key it needs is
 *  absent (AvatarModel: at()).  `only` (optional): parse just these member names. */
inline std::map<std::string, Array> load(const std::string& path, const std::vector<std::string>* only = nullptr) {
    std::ifstream ifs(path, std::ios::binary);
    if (!ifs) throw std::runtime_error("npz: cannot open " + path);
    std::vector<unsigned char> z((std::istreambuf_iterator<char>(ifs)), std::istreambuf_iterator<char>());
    auto need = [&](size_t o, size_t len) { if (o > z.size() || len > z.size() - o) throw std::runtime_error("npz: truncated or corrupt zip structure in " + path); };
    auto u16 = [&](size_t o) { need(o, 2); return (uint32_t)z[o] | ((uint32_t)z[o + 1] << 8); };
    auto u32 = [&](size_t o) { need(o, 4); return (uint32_t)z[o] | ((uint32_t)z[o + 1] << 8) | ((uint32_t)z[o + 2] << 16) | ((uint32_t)z[o + 3] << 24); };
    // end-of-central-directory record
    if (z.size() < 22) throw std::runtime_error("npz: not a zip file");
    size_t eocd = z.size() - 22;
    while (eocd > 0 && u32(eocd) != 0x06054b50) --eocd;
    if (u32(eocd) != 0x06054b50) throw std::runtime_error("npz: end of central directory not found");
    const size_t nent = u16(eocd + 10);
    size_t cd = u32(eocd + 16);
    std::map<std::string, Array> out;
    for (size_t e = 0; e < nent; ++e) {
        if (u32(cd) != 0x02014b50) throw std::runtime_error("npz: bad central directory entry");
        const uint32_t method = u16(cd + 10);
        size_t csize = u32(cd + 20), usize = u32(cd + 24);
        const size_t nlen = u16(cd + 28), xlen = u16(cd + 30), clen = u16(cd + 32);
        size_t lho = u32(cd + 42);
        need(cd + 46, nlen + xlen + clen);
        std::string name((const char*)z.data() + cd + 46, nlen);
        // zip64 extra field (numpy writes it for large members)
        size_t xo = cd + 46 + nlen;
        const size_t xend = xo + xlen;
        while (xo + 4 <= xend) {
            const uint32_t id = u16(xo), sz = u16(xo + 2);
            if (id == 0x0001) {
                size_t o = xo + 4;
                auto u64 = [&](size_t oo) { need(oo, 8); uint64_t v; std::memcpy(&v, z.data() + oo, 8); return (size_t)v; };
                if (usize == 0xffffffffu) { usize = u64(o); o += 8; }
                if (csize == 0xffffffffu) { csize = u64(o); o += 8; }
                if (lho == 0xffffffffu) { lho = u64(o); o += 8; }
            }
            xo += 4 + sz;
        }
        if (name.size() > 4 && name.substr(name.size() - 4) == ".npy") name = name.substr(0, name.size() - 4);
        const size_t next_cd = cd + 46 + nlen + xlen + clen;
        if (only) {
            bool wanted = false;
            for (const std::string& k : *only) wanted = wanted || k == name;
            if (!wanted) { cd = next_cd; continue; }
        }
        need(lho, 30);
        const size_t dn = u16(lho + 26), dx = u16(lho + 28);
        need(lho + 30 + dn + dx, csize);
        if (usize > ((size_t)1 << 36)) throw std::runtime_error("npz: implausible member size for " + name);
        if (method == 8 && (csize > 0xffffffffu || usize > 0xffffffffu)) throw std::runtime_error("npz: deflated member above 4 GiB not supported: " + name);
        const unsigned char* src = z.data() + lho + 30 + dn + dx;
        std::vector<unsigned char> raw(usize);
        if (method == 0) {
            if (usize > csize) throw std::runtime_error("npz: stored member shorter than declared: " + name);
            std::memcpy(raw.data(), src, usize);
        } else if (method == 8) {
            z_stream zs;
            std::memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -MAX_WBITS) != Z_OK) throw std::runtime_error("npz: inflateInit failed");
            zs.next_in = const_cast<unsigned char*>(src); zs.avail_in = (uInt)csize;
            zs.next_out = raw.data(); zs.avail_out = (uInt)usize;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END) throw std::runtime_error("npz: inflate failed for " + name);
        } else {
            throw std::runtime_error("npz: unsupported compression method");
        }
        try {
            out[name] = parse_npy(raw);
        } catch (const std::exception& e) {          // not a numeric array (incl. std::stoull / std::stoi logic errors on an odd
                                                     // header): skipped unless the caller asked for exactly this member
            if (only) throw std::runtime_error(std::string(e.what()) + " (member " + name + ")");
        }
        cd = next_cd;
    }
    return out;
}

}  // namespace npz
}  // namespace ark
