// cnpy_ref.cpp — drives the REFERENCE's own .npz reader (cnpy::npz_load, /root/reference/cnpy.cpp:246-300, compiled from the
// reference tree where it lies by oracle/Makefile into oracle/_ref/libcnpy_ref.so; nothing of it is copied here) so that
// include/ark/Npz.h - the product's independent reader for SURVEY.md §8 row f2 - can be checked against it byte for byte.
// TEST INFRASTRUCTURE ONLY: loaded by tests/test_npz_against_cnpy.py, never by the product.
#include <cstring>
#include <string>
#include <vector>

#include "cnpy.h"

namespace {
struct Handle {
    cnpy::npz_t z;
    std::vector<std::string> names;
};
}  // namespace

extern "C" {

void* cnpyref_open(const char* path) {
    try {
        Handle* h = new Handle();
        h->z = cnpy::npz_load(path);
        for (auto& kv : h->z) h->names.push_back(kv.first);
        return h;
    } catch (...) { return nullptr; }
}
void cnpyref_close(void* hv) { delete (Handle*)hv; }
int cnpyref_count(void* hv) { return (int)((Handle*)hv)->names.size(); }
const char* cnpyref_name(void* hv, int i) { return ((Handle*)hv)->names[i].c_str(); }
// word size, fortran flag, number of dimensions and shape (up to 8) of member i; returns the number of data bytes
long long cnpyref_info(void* hv, int i, int* word_size, int* fortran_order, int* ndim, long long* shape8) {
    Handle* h = (Handle*)hv;
    const cnpy::NpyArray& a = h->z.at(h->names[i]);
    *word_size = (int)a.word_size; *fortran_order = a.fortran_order ? 1 : 0; *ndim = (int)a.shape.size();
    for (size_t k = 0; k < a.shape.size() && k < 8; ++k) shape8[k] = (long long)a.shape[k];
    return (long long)a.num_bytes();
}
void cnpyref_bytes(void* hv, int i, unsigned char* out) {
    Handle* h = (Handle*)hv;
    const cnpy::NpyArray& a = h->z.at(h->names[i]);
    if (a.num_bytes()) std::memcpy(out, a.data<unsigned char>(), a.num_bytes());
}

}  // extern "C"
