// jumanpp_gpu_train: the reference's `jumanpp_v2_train` (src/jumandic/main/jumanpp_train.cc) with the analysis of the
// training examples on the device.  Same flags where the function exists here; the output is a .jppmdl with the
// trained perceptron part appended, loadable by the reference's analyser and by jumanpp_gpu.
//   not here: --rnn-model (embedding an RNN into a model is an offline repack, not the analysis path),
//   --partial-corpus, --scw-dump-dir, --threads (accepted and
//   ignored: the device analyses the whole batch at once).
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "model_image.h"
#include "train/train_env.h"

using namespace jumanpp_amd;

namespace {
bool flagValue(const std::string& arg, const char* name, std::string* out) {
  const std::string pfx = std::string("--") + name + "=";
  if (arg.compare(0, pfx.size(), pfx) != 0) return false;
  *out = arg.substr(pfx.size());
  return true;
}
void usage() {
  std::cout << "jumanpp_gpu_train --model-input=FILE --model-output=FILE --corpus=FILE [--size=15] [--seed=N]\n"
               "  [--training-mode=full|falloff|violation] [--scw-c=1] [--scw-phi=5] [--beam=5] [--batch=1]\n"
               "  [--max-batch-iters=1] [--max-epochs=1] [--epsilon=1e-3] [--corpus-format=morph|csv]\n"
               "  [--gb-left-min=N --gb-left-max=N [--gb-right-min=N --gb-right-max=N --gb-rcheck-min=N --gb-rcheck-max=N] [--gb-first-full]]\n"
               "  [--corpus-comment=TEXT] [--device=0]\n"
               "\n"
               "--batch=1 writes the model file jumanpp_v2_train writes (byte for byte).  With --batch=N > 1 all N examples of\n"
               "a batch are analysed with the weights the batch started with and the N SCW updates are applied afterwards;\n"
               "the reference interleaves each example's analysis with the updates of the examples before it\n"
               "(trainOneBatch / handleProcessedTrainer), so the two trainers' models DIFFER for any batch above 1.\n";
}
}  // namespace

int main(int argc, char** argv) {
  train::TrainingArguments a;
  for (int i = 1; i < argc; ++i) {
    const std::string arg = argv[i];
    std::string v;
    if (flagValue(arg, "model-input", &v)) a.modelFilename = v;
    else if (flagValue(arg, "model-output", &v)) a.outputFilename = v;
    else if (flagValue(arg, "corpus", &v)) a.corpusFilename = v;
    else if (flagValue(arg, "corpus-comment", &v)) a.comment = v;
    else if (flagValue(arg, "size", &v)) a.sizeExponent = (uint32_t)std::strtoul(v.c_str(), nullptr, 0);
    else if (flagValue(arg, "seed", &v)) a.randomSeed = (uint32_t)std::strtoul(v.c_str(), nullptr, 0);
    else if (flagValue(arg, "training-mode", &v)) {
      if (v == "full") a.mode = train::TrainingMode::Full;
      else if (v == "falloff") a.mode = train::TrainingMode::FalloffBeam;
      else if (v == "violation") a.mode = train::TrainingMode::MaxViolation;
      else {
        std::cerr << "unknown training mode: " << v << "\n";
        return 1;
      }
    } else if (flagValue(arg, "corpus-format", &v)) {
      if (v == "morph") a.inputFormat = train::CorpusFormat::Morph;
      else if (v == "csv") a.inputFormat = train::CorpusFormat::Csv;
      else {
        std::cerr << "unknown corpus format: " << v << "\n";
        return 1;
      }
    } else if (flagValue(arg, "scw-c", &v)) a.scw.C = std::strtof(v.c_str(), nullptr);
    else if (flagValue(arg, "scw-phi", &v)) a.scw.phi = std::strtof(v.c_str(), nullptr);
    else if (flagValue(arg, "beam", &v)) a.beamSize = std::atoi(v.c_str());
    else if (flagValue(arg, "batch", &v)) a.batchSize = (uint32_t)std::strtoul(v.c_str(), nullptr, 0);
    else if (flagValue(arg, "threads", &v)) (void)v;
    else if (flagValue(arg, "max-batch-iters", &v)) a.batchMaxIterations = (uint32_t)std::strtoul(v.c_str(), nullptr, 0);
    else if (flagValue(arg, "max-epochs", &v)) a.maxEpochs = (uint32_t)std::strtoul(v.c_str(), nullptr, 0);
    else if (flagValue(arg, "epsilon", &v)) a.batchLossEpsilon = std::strtof(v.c_str(), nullptr);
    else if (flagValue(arg, "gb-left-min", &v)) a.globalBeam.minLeftBeam = std::atoi(v.c_str());
    else if (flagValue(arg, "gb-left-max", &v)) a.globalBeam.maxLeftBeam = std::atoi(v.c_str());
    else if (flagValue(arg, "gb-right-min", &v)) a.globalBeam.minRightBeam = std::atoi(v.c_str());
    else if (flagValue(arg, "gb-right-max", &v)) a.globalBeam.maxRightBeam = std::atoi(v.c_str());
    else if (flagValue(arg, "gb-rcheck-min", &v)) a.globalBeam.minRightCheck = std::atoi(v.c_str());
    else if (flagValue(arg, "gb-rcheck-max", &v)) a.globalBeam.maxRightCheck = std::atoi(v.c_str());
    else if (arg == "--gb-first-full") a.globalBeam.fullFirstIter = true;
    else if (flagValue(arg, "device", &v)) a.device = std::atoi(v.c_str());
    else if (arg == "-h" || arg == "--help") {
      usage();
      return 0;
    } else {
      std::cerr << "unknown argument: " << arg << "\n";
      usage();
      return 1;
    }
  }
  if (a.modelFilename.empty() || a.corpusFilename.empty() || a.outputFilename.empty()) {
    std::cerr << "Model, corpus or output filename was not specified\n";
    usage();
    return 1;
  }
  ModelImage model;
  Status s = model.loadModelForTraining(a.modelFilename);
  if (!s) {
    std::cerr << "failed to read model from disk: " << s << "\n";
    return 1;
  }
  train::TrainingEnv env;
  s = env.initialize(a, &model);
  if (!s) {
    std::cerr << "failed to initialize training process: " << s << "\n";
    return 1;
  }
  s = env.loadInput(a.corpusFilename);
  if (!s) {
    std::cerr << "failed to open corpus filename: " << s << "\n";
    return 1;
  }
  s = train::trainModel(&env, a);
  if (!s) {
    std::cerr << "failed to train: " << s << "\n";
    return 1;
  }
  std::cerr << "trained on " << env.examplesSeen() << " example passes, " << env.goldNodesAdded() << " gold nodes added, last epoch loss="
            << env.epochLoss() << "\n";
  const double* ms = env.stageMs();
  std::cerr << "stage ms: analyse+seed hook " << ms[0] << ", lattice fetch " << ms[1] << ", n-gram read-outs " << ms[2] << ", gold scores + loss + feature diff " << ms[3]
            << ", SCW updates " << ms[4] << ", weights upload " << ms[5] << "\n";
  s = model.saveWithPerceptron(a.outputFilename, env.scw().weights().data(), env.scw().exponent(), a.comment);
  if (!s) {
    std::cerr << "failed to save model: " << s << "\n";
    return 1;
  }
  return 0;
}
