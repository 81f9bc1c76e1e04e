// Test infrastructure (not product): loads each file given on the command line through Hdf5Model and NifModel's PTNIF
// path exactly as the host does.  Built with -fsanitize=address,undefined by tests/test_hdf5.py: a malformed file must
// end in a std::exception (counted), never in a sanitizer report (the process aborts with a non-zero status).
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>

#include "Hdf5Model.hpp"
#include "NifModel.hpp"

int main(int argc, char** argv) {
  int loaded = 0, rejected = 0;
  const std::string meta = argv[1];
  for (int i = 2; i < argc; ++i) {
    try {
      const std::string f = argv[i];
      if (f.size() > 6 && f.compare(f.size() - 6, 6, ".ptnif") == 0) {
        NifModel::Data d(f, meta);
        loaded += (int)!d.getLayers().empty();
      } else {
        Hdf5Model m(f);
        std::size_t bytes = 0;
        for (const auto& l : m.get()) bytes += l.kernelData.storage.size() + l.biasData.storage.size();
        loaded += bytes > 0;
      }
    } catch (const std::exception&) {
      rejected += 1;
    }
  }
  std::printf("loaded %d rejected %d\n", loaded, rejected);
  return 0;
}
