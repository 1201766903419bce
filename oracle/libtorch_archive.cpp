// Test infrastructure (never linked into or called by the product): drives libtorch's own archive code the way tch 0.16's
// torch-sys does for `VarStore::save` / `VarStore::load` on non-safetensors paths (the reference's `*.pt.tch` files,
// border-tch-agent/src/dqn/base.rs:348-362), so that border_amd/csrc/tch_archive.hpp can be pinned against the real
// serializer instead of a description of it.  tch is a third-party dependency absent from /root/reference; the two call
// sequences below restate torch-sys `libtch/torch_api.cpp` (`at_save_multi`, `at_load_callback`) from its published source.
//
//   libtorch_archive write <file> <name> <d0,d1,..> [<name> <dims> ...]
//        builds f32 tensors with the deterministic fill value(name_index, element) = sin(0.37 * (element + 1) + index)
//        and saves them:   OutputArchive ar; ar.write(name, tensor); ar.save_to(file);
//   libtorch_archive read <file>
//        loads with torch::jit::load(file) and prints, for each named_parameters() entry,
//        "<name> <ndim> <dims...> <fnv1a64 of the raw f32 bytes>"
//
// Built against the libtorch inside the installed PyTorch wheel by oracle/build_libtorch_archive.sh.
#include <torch/script.h>
#include <torch/serialize/archive.h>

#include <cmath>
#include <cstdio>
#include <sstream>
#include <string>
#include <vector>

static uint64_t fnv1a(const void* p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= ((const unsigned char*)p)[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv)
{
    if (argc >= 3 && std::string(argv[1]) == "write") {
        torch::serialize::OutputArchive archive;
        for (int a = 3, index = 0; a + 1 < argc; a += 2, ++index) {
            std::vector<int64_t> dims;
            std::stringstream ss(argv[a + 1]);
            for (std::string tok; std::getline(ss, tok, ',');) if (!tok.empty()) dims.push_back(std::stoll(tok));
            torch::Tensor t = torch::empty(dims, torch::kFloat32);
            float* d = t.data_ptr<float>();
            for (int64_t e = 0; e < t.numel(); ++e) d[e] = (float)std::sin(0.37 * (double)(e + 1) + (double)index);
            archive.write(argv[a], t, /*is_buffer=*/false);
        }
        archive.save_to(std::string(argv[2]));
        return 0;
    }
    if (argc == 3 && std::string(argv[1]) == "read") {
        torch::jit::Module m = torch::jit::load(std::string(argv[2]));
        for (const auto& p : m.named_parameters()) {
            torch::Tensor t = p.value.contiguous().to(torch::kFloat32);
            std::printf("%s %d", p.name.c_str(), (int)t.dim());
            for (auto s : t.sizes()) std::printf(" %lld", (long long)s);
            std::printf(" %016llx\n", (unsigned long long)fnv1a(t.data_ptr<float>(), (size_t)t.numel() * 4));
        }
        return 0;
    }
    std::fprintf(stderr, "usage: libtorch_archive write <file> <name> <dims> ... | read <file>\n");
    return 2;
}
