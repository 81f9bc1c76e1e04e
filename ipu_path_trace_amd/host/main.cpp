// ipu_trace for MI355X: CLI of the reference (src/main.cpp:8-102, PathTracerApp.cpp:794-830) without Boost.
#include <cstdlib>
#include <iostream>
#include <stdexcept>

#include "PathTracerApp.hpp"
#include "logging.hpp"

static OptionMap parseOptions(int argc, char** argv, const std::vector<OptionSpec>& specs) {
  OptionMap vm;
  for (auto& s : specs) if (!s.required && !(s.name == "help")) vm.values[s.name] = s.defaultValue;
  auto find = [&](const std::string& tok) -> const OptionSpec* {
    for (auto& s : specs) {
      if (tok == "--" + s.name) return &s;
      if (s.shortName && tok == std::string("-") + s.shortName) return &s;
    }
    return nullptr;
  };
  for (int i = 1; i < argc; ++i) {
    std::string tok = argv[i], val;
    bool inlineVal = false;
    auto eq = tok.find('=');
    if (tok.rfind("--", 0) == 0 && eq != std::string::npos) { val = tok.substr(eq + 1); tok = tok.substr(0, eq); inlineVal = true; }
    const OptionSpec* s = find(tok);
    if (!s) throw std::runtime_error("unrecognised option '" + tok + "'");
    if (s->isSwitch) { vm.values[s->name] = "true"; continue; }
    if (!inlineVal) {
      if (i + 1 >= argc) throw std::runtime_error("the required argument for option '--" + s->name + "' is missing");
      val = argv[++i];
    }
    vm.values[s->name] = val;
  }
  if (vm.has("help")) {
    std::cout << "Options:\n";
    for (auto& s : specs) {
      std::cout << "  --" << s.name;
      if (s.shortName) std::cout << " [ -" << s.shortName << " ]";
      if (!s.isSwitch) std::cout << " arg" << (s.defaultValue.empty() ? "" : " (=" + s.defaultValue + ")");
      std::cout << "  " << s.help << "\n";
    }
    throw std::runtime_error("Show help");
  }
  for (auto& s : specs) if (s.required && !vm.has(s.name)) throw std::runtime_error("the option '--" + s.name + "' is required but missing");
  if (!vm.str("save-exe").empty() && !vm.str("load-exe").empty())
    throw std::logic_error("You can not set both save-exe and load-exe.");   // main.cpp:63-66
  return vm;
}

int main(int argc, char** argv) {
  try {
    PathTracerApp app;
    auto specs = PathTracerApp::addToolOptions();
    auto opts = parseOptions(argc, argv, specs);
    pt_log::setLevel(opts.str("log-level"));
    app.init(opts);
    if (opts.flag("compile-only")) {
      // The reference builds and compiles its graph (validating options and the NIF on the way), optionally saves the
      // executable and stops without attaching to a device (ipu_utils.hpp:523-526, PathTracerApp.cpp:91-92).  There is no
      // graph to compile here: options and assets have been validated by init(); nothing is rendered.
      pt_log::info_("Compile only mode selected: finished.");
      return EXIT_SUCCESS;
    }
    app.attach();
    app.execute();
    return EXIT_SUCCESS;
  } catch (const std::exception& e) {
    // GraphManager::run catches once, logs and returns EXIT_FAILURE (ipu_utils.hpp:532-535)
    pt_log::error_("Exception: {}", e.what());
    return EXIT_FAILURE;
  }
}
