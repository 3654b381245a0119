// TEST INFRASTRUCTURE ONLY (oracle build).
// Drives the reference's own static-feature code generator
// (src/core/codegen/feature_codegen.h:22-33) for the jumandic spec
// (src/jumandic/shared/jumandic_spec.cc:31) so that oracle/_ref contains the
// exact generated scoring code the reference CLI is built with
// (src/CMakeLists.txt jpp_feature_codegen, src/jumandic/CMakeLists.txt).
// usage: jumandic_codegen <base-filename> <class-name> <out-dir>
#include <iostream>

#include "core/codegen/feature_codegen.h"
#include "core/spec/spec_dsl.h"
#include "jumandic/shared/jumandic_spec.h"

int main(int argc, char** argv) {
  if (argc != 4) {
    std::cerr << "usage: " << argv[0] << " base class outdir\n";
    return 2;
  }
  namespace cg = jumanpp::core::features::codegen;
  cg::FeatureCodegenConfig conf;
  conf.filename = argv[1];
  conf.className = argv[2];
  conf.baseDirectory = argv[3];

  jumanpp::core::spec::AnalysisSpec spec;
  auto st = jumanpp::jumandic::SpecFactory::makeSpec(&spec);
  if (!st) {
    std::cerr << "spec build failed: " << st << "\n";
    return 1;
  }
  cg::StaticFeatureCodegen gen{conf, spec};
  st = gen.generateAndWrite();
  if (!st) {
    std::cerr << "codegen failed: " << st << "\n";
    return 1;
  }
  return 0;
}
