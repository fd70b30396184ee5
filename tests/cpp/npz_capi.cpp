// npz_capi.cpp — a C view of include/ark/Npz.h (the product's .npz reader) for tests/test_npz_against_cnpy.py.
#include <cstring>
#include <string>
#include <vector>

#include "ark/Npz.h"

namespace {
struct Handle {
    std::map<std::string, ark::npz::Array> z;
    std::vector<std::string> names;
    std::string error;
};
}  // namespace

extern "C" {

void* arknpz_open(const char* path) {
    Handle* h = new Handle();
    try {
        h->z = ark::npz::load(path);
        for (auto& kv : h->z) h->names.push_back(kv.first);
    } catch (const std::exception& e) { h->error = e.what(); }
    return h;
}
const char* arknpz_error(void* hv) { return ((Handle*)hv)->error.c_str(); }
void arknpz_close(void* hv) { delete (Handle*)hv; }
int arknpz_count(void* hv) { return (int)((Handle*)hv)->names.size(); }
const char* arknpz_name(void* hv, int i) { return ((Handle*)hv)->names[i].c_str(); }
long long arknpz_info(void* hv, int i, int* is_int, int* ndim, long long* shape8) {
    Handle* h = (Handle*)hv;
    const ark::npz::Array& a = h->z.at(h->names[i]);
    *is_int = a.is_int ? 1 : 0; *ndim = (int)a.shape.size();
    for (size_t k = 0; k < a.shape.size() && k < 8; ++k) shape8[k] = (long long)a.shape[k];
    return (long long)a.size();
}
// values in logical C order, as doubles (floats) or long longs (integers)
void arknpz_values(void* hv, int i, double* f_out, long long* i_out) {
    Handle* h = (Handle*)hv;
    const ark::npz::Array& a = h->z.at(h->names[i]);
    if (a.is_int) { for (size_t k = 0; k < a.i.size(); ++k) i_out[k] = a.i[k]; }
    else { for (size_t k = 0; k < a.f.size(); ++k) f_out[k] = a.f[k]; }
}

}  // extern "C"
