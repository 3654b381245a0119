// jumanpp_gpu: the reference CLI's read-analyze-format loop
// (src/jumandic/main/jumanpp.cc:100-182, PlainStreamReader
// src/core/input/stream_reader.cc:12-38) with the per-line Analyzer::analyze
// replaced by batched GpuAnalyzer::analyzeBatch.  Output is byte-identical to
// `jumanpp_v2 --model=... ` in the JUMAN format.
//
// usage: jumanpp_gpu --model=MODEL.jppmdl [--beam=5] [--global-beam=6] [--right-check=1]
//                    [-c CONFIG] [--rnn-nce-bias=X --rnn-unk-constant=X --rnn-unk-length=X
//                    --feature-weight-perceptron=X --feature-weight-rnn=X]
//                    [--right-beam=5] [--no-rnn] [-s N | -M | -F | --segment | --dic-subset] [--partial-input]
//                    [--auto-nbest=B:S:M] [--batch=65536] [--threads=N] [--no-pipeline]
//                    [--device=0 | --devices=0-7] [--timing] [-o OUT] [INPUT...]
//
// --devices=LIST (e.g. 0-7 or 0,2,5) replaces the reference's one-thread loop over sentences
// (jumanpp.cc:156-179) by one analysis thread per GPU: batches are dealt to the devices in turn, every
// device has its own pair of analyzers, and the formatter consumes the batches in input order, so the
// output is the same byte string for any device list.  Sentences are independent: no collective.
//
// Files in, file out (INPUT... and -o OUT, the bulk case) takes the SHARDED pipeline: no stage is shared between
// devices.  The inputs are mapped; one scanner cuts them into batches of whole examples at newline boundaries
// (memchr, ~10 GB/s) and deals them to the devices; every device has its own thread that splits its batch into
// lines and analyses it, its own format workers, and its own writer, which pwrite()s the batch at the offset a
// tiny sequencer hands out in input order (the prefix sum of the formatted sizes).  Everything else (stdin,
// stdout, --partial-input, --no-pipeline) keeps the general four-stage pipeline below.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <future>
#include <thread>

#include <sched.h>
#include <vector>

#include "derived_cache.h"
#include "format_table.h"
#include "gpu_analyzer.h"
#include "juman_format.h"
#include "lattice_format.h"
#include "rnn_external.h"
#include "simple_formats.h"

using namespace jumanpp_amd;

namespace {

struct Conf {
  std::string model;
  int beam = 5, globalBeam = 6, rightCheck = 1, rightBeam = 5;  // jumanpp_args.h:50-54
  bool noRnn = false;
  size_t batch = 65536;
  bool batchGiven = false;    // --batch on the command line or in the config file: taken as it is
  int device = 0;
  std::vector<int> devices;   // --devices=LIST: one analysis thread (and analyzer pair) per listed GPU
  std::string output;
  std::vector<std::string> inputs;
  bool timing = false;
  int outputShards = 1;        // --output-shards=K: the input in K contiguous parts, part k's output in OUT.partNNNN (file to file only)
  bool cleanExit = false;      // --clean-exit: run every destructor at the end of a file-to-file run instead of leaving at once
  bool noImageCache = false;   // --no-image-cache: derive the T0 records and the format table afresh, write no cache
  bool noReserve = false;   // --no-reserve: the analyzers size their buffers batch by batch (round-4 behaviour)
  bool help = false;
  int lattice = 0;  // -s N / --lattice N / --specifics N: LatticeFormat with the N best paths; -1 = beam width
  enum { Juman, Morph, FullMorph, Segment, DicSubset } kind = Juman;
  std::string segmentSeparator = " ";
  bool partialInput = false;  // --partial-input: InputType::PartiallyAnnotated
  int autoStep = 0;           // --auto-nbest=base:step:max (jumanpp_args.cc:270-279)
  std::string configFile;     // -c / --config: file of whitespace-separated arguments, read before the command line
  std::string rnnModel;       // --rnn-model: a separate faster-rnnlm model (PATH + PATH.nnet) instead of an embedded one
  ExternalRnnConfig rnnNames; // --rnn-fields, --rnn-separator, --rnn-unk, --rnn-eos
  RnnConfigOverride rnn;  // --rnn-nce-bias, --rnn-unk-constant, --rnn-unk-length, --feature-weight-*
  int threads = 0;            // --threads=N format workers (0: one per hardware thread, at most 32)
  bool pipeline = true;       // --no-pipeline: one analyzer, read/analyse/format strictly in turn per batch
  int pipelinesPerDevice = 2;  // --pipelines-per-device=N (bulk runs)
  bool hostFormat = false;    // --host-format: JUMAN text from the host formatters even where the device can print it
};

bool argValue(int argc, const char** argv, int& i, const char* name, std::string* out) {
  size_t n = std::strlen(name);
  const char* a = argv[i];
  if (std::strncmp(a, name, n) != 0) return false;
  if (a[n] == '=') {
    *out = a + n + 1;
    return true;
  }
  if (a[n] == 0 && i + 1 < argc) {
    *out = argv[++i];
    return true;
  }
  if (n == 2 && a[n] != 0) {  // -s5
    *out = a + n;
    return true;
  }
  return false;
}

struct Example {
  std::string comment;
  std::string input;
  Status readStatus;
  std::unique_ptr<PartialExample> partial;  // --partial-input only
};

// one batch on its way through read -> analyse -> format
struct Job {
  std::vector<Example> batch;
  int device = 0;    // index into the device list
  int analyzer = 0;  // which of that device's analyzers
  Status batchStatus;
  double gpuMs = 0;
};

// the text of one formatted batch, in chunks of consecutive sentences
struct Formatted {
  std::vector<std::string> text, errors;
};

// SHARDED pipeline: one batch = a run of whole examples of a mapped input file
struct ShardJob {
  size_t seq = 0;              // position in input order (within its output shard)
  int shard = 0;               // --output-shards: which part of the input, = which output file
  const char* data = nullptr;  // the batch's lines, each with its newline (the last one of a file maybe without)
  size_t bytes = 0;
  std::vector<StringPiece> inputs, comments;   // per example, pointing into the mapping
  std::vector<std::pair<size_t, Status>> readErrors;   // (example, status): over-long inputs / comments, sorted
  int device = 0, analyzer = 0;
  Status batchStatus;
  double gpuMs = 0;
  bool lastReadOk = true;
  std::vector<std::string> text, errors;   // formatted chunks of consecutive sentences
  // device text: the batch's bytes stay in the result's (page-locked) host copy; the writer gets them as segments
  TextBatch deviceText;
  std::vector<struct iovec> segments;
  size_t outBytes = 0;
  uint64_t outOffset = 0;
};

struct MappedFile {
  const char* data = nullptr;
  size_t size = 0;
  bool open(const std::string& path) {
    int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
      ::close(fd);
      return false;
    }
    size = (size_t)st.st_size;
    if (size) {
      void* p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (p == MAP_FAILED) {
        ::close(fd);
        return false;
      }
      madvise(p, size, MADV_SEQUENTIAL);
      data = static_cast<const char*>(p);
    }
    ::close(fd);
    return true;
  }
  ~MappedFile() {
    if (data) munmap(const_cast<char*>(data), size);
  }
};

// a fixed set of workers that run `fn(worker)` together, once per call of run()
class WorkerPool {
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  std::function<void(int)> fn_;
  size_t generation_ = 0, running_ = 0;
  bool stop_ = false;

 public:
  explicit WorkerPool(int n) {
    for (int t = 0; t < n; ++t)
      threads_.emplace_back([this, t]() {
        size_t seen = 0;
        for (;;) {
          std::function<void(int)> fn;
          {
            std::unique_lock<std::mutex> l(mu_);
            cv_.wait(l, [&] { return stop_ || generation_ != seen; });
            if (stop_) return;
            seen = generation_;
            fn = fn_;
          }
          fn(t);
          std::lock_guard<std::mutex> l(mu_);
          if (--running_ == 0) done_.notify_all();
        }
      });
  }
  void run(std::function<void(int)> fn) {
    std::unique_lock<std::mutex> l(mu_);
    fn_ = std::move(fn);
    running_ = threads_.size();
    ++generation_;
    cv_.notify_all();
    done_.wait(l, [&] { return running_ == 0; });
  }
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> l(mu_);
      stop_ = true;
      cv_.notify_all();
    }
    for (auto& t : threads_) t.join();
  }
};

std::string statusText(const Status& s) {
  std::ostringstream o;
  o << s;
  return o.str();
}

// Hardware threads this process may really use: the affinity mask and the cgroup CPU quota, not what is visible (a
// container that sees 256 CPUs and is granted 16 is throttled for the rest of the scheduler period once its threads
// have used the quota up: 32 format workers there stalled the analysis thread for tens of milliseconds at a time).
unsigned usableCores() {
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) {
    const int c = CPU_COUNT(&set);
    if (c > 0) n = std::min(n, (unsigned)c);
  }
  {
    std::ifstream f("/sys/fs/cgroup/cpu.max");   // cgroup v2: "<quota|max> <period>"
    std::string quota;
    long long period = 0;
    if (f >> quota >> period && quota != "max" && period > 0) {
      const long long q = std::atoll(quota.c_str());
      if (q > 0) n = std::min(n, (unsigned)std::max(1LL, q / period));
    }
  }
  {
    std::ifstream fq("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), fp("/sys/fs/cgroup/cpu/cpu.cfs_period_us");   // cgroup v1
    long long q = -1, per = 0;
    if (fq >> q && fp >> per && q > 0 && per > 0) n = std::min(n, (unsigned)std::max(1LL, q / per));
  }
  return n;
}

template <typename T>
class BoundedQueue {
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<T> items_;
  size_t cap_;
  bool closed_ = false;

 public:
  explicit BoundedQueue(size_t cap) : cap_(cap) {}
  void push(T v) {
    std::unique_lock<std::mutex> l(mu_);
    cv_.wait(l, [&] { return items_.size() < cap_; });
    items_.push_back(std::move(v));
    cv_.notify_all();
  }
  // false once the queue is closed and drained
  bool pop(T* out) {
    std::unique_lock<std::mutex> l(mu_);
    cv_.wait(l, [&] { return !items_.empty() || closed_; });
    if (items_.empty()) return false;
    *out = std::move(items_.front());
    items_.pop_front();
    cv_.notify_all();
    return true;
  }
  void close() {
    std::lock_guard<std::mutex> l(mu_);
    closed_ = true;
    cv_.notify_all();
  }
};

class Semaphore {
  std::mutex mu_;
  std::condition_variable cv_;
  int count_;

 public:
  explicit Semaphore(int n) : count_(n) {}
  void acquire() {
    std::unique_lock<std::mutex> l(mu_);
    cv_.wait(l, [&] { return count_ > 0; });
    --count_;
  }
  void release() {
    std::lock_guard<std::mutex> l(mu_);
    ++count_;
    cv_.notify_one();
  }
};

struct Clock {
  std::chrono::steady_clock::time_point start = std::chrono::steady_clock::now();
  double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - start).count(); }
};

// one list of arguments (a config file's tokens or the command line without argv[0]) into `conf`;
// later lists override earlier ones, like JumanppConf::mergeWith (jumanpp_args.cc:343-400)
bool parseArgList(const std::vector<std::string>& args, Conf& conf) {
  std::vector<const char*> ptrs;
  for (auto& a : args) ptrs.push_back(a.c_str());
  const int argc = (int)ptrs.size();
  const char** argv = ptrs.data();
  for (int i = 0; i < argc; ++i) {
    std::string v;
    if (argValue(argc, argv, i, "--model", &v)) conf.model = v;
    else if (argValue(argc, argv, i, "--beam", &v)) conf.beam = std::atoi(v.c_str());
    else if (argValue(argc, argv, i, "--global-beam", &v)) conf.globalBeam = std::atoi(v.c_str());
    else if (argValue(argc, argv, i, "--right-check", &v)) conf.rightCheck = std::atoi(v.c_str());
    else if (argValue(argc, argv, i, "--right-beam", &v)) conf.rightBeam = std::atoi(v.c_str());
    else if (argValue(argc, argv, i, "--batch", &v)) {
      conf.batch = (size_t)std::atoll(v.c_str());
      conf.batchGiven = true;
    }
    else if (argValue(argc, argv, i, "--devices", &v)) {
      // comma-separated ordinals and ranges: 0-7, 0,2,5, 0-3,6
      conf.devices.clear();
      for (size_t p = 0; p <= v.size();) {
        size_t e = v.find(',', p);
        if (e == std::string::npos) e = v.size();
        std::string part = v.substr(p, e - p);
        size_t dash = part.find('-');
        int lo = std::atoi(part.c_str()), hi = dash == std::string::npos ? lo : std::atoi(part.c_str() + dash + 1);
        bool digits = !part.empty() && part.find_first_not_of("0123456789-") == std::string::npos && part[0] != '-' &&
                      part[part.size() - 1] != '-' && (dash == std::string::npos || part.find('-', dash + 1) == std::string::npos);
        if (!digits || lo < 0 || hi < lo || hi - lo > 1023) {
          std::cerr << "bad device list " << v << "\n";
          return false;
        }
        for (int d = lo; d <= hi; ++d) conf.devices.push_back(d);
        p = e + 1;
      }
    }
    else if (argValue(argc, argv, i, "--device", &v)) conf.device = std::atoi(v.c_str());
    else if (argValue(argc, argv, i, "--output", &v) || argValue(argc, argv, i, "-o", &v)) conf.output = v;
    else if (argValue(argc, argv, i, "--lattice", &v) || argValue(argc, argv, i, "--specifics", &v) ||
             argValue(argc, argv, i, "-s", &v) || argValue(argc, argv, i, "-L", &v)) conf.lattice = std::atoi(v.c_str());
    else if (argValue(argc, argv, i, "--segment-separator", &v)) conf.segmentSeparator = v;
    else if (std::strcmp(argv[i], "--segment") == 0) conf.kind = Conf::Segment;
    else if (std::strcmp(argv[i], "--dic-subset") == 0) conf.kind = Conf::DicSubset;
    else if (argValue(argc, argv, i, "--format", &v)) {  // format_map(), jumanpp_args.cc:48-63 (no protobuf formats here)
      conf.lattice = 0;
      if (v == "juman") conf.kind = Conf::Juman;
      else if (v == "segment") conf.kind = Conf::Segment;
      else if (v == "morph") conf.kind = Conf::Morph;
      else if (v == "full-morph") conf.kind = Conf::FullMorph;
      else if (v == "dic-subset") conf.kind = Conf::DicSubset;
      else if (v == "lattice") conf.lattice = 1;  // beamOutput keeps its default of 1 (jumanpp_args.h:51)
      else {
        std::cerr << "unknown output format " << v << " (juman, segment, morph, full-morph, dic-subset, lattice)\n";
        return false;
      }
    }
    else if (std::strcmp(argv[i], "--morph") == 0 || std::strcmp(argv[i], "-M") == 0) conf.kind = Conf::Morph;
    else if (std::strcmp(argv[i], "--full-morph") == 0 || std::strcmp(argv[i], "-F") == 0) conf.kind = Conf::FullMorph;
    else if (std::strcmp(argv[i], "--juman") == 0 || std::strcmp(argv[i], "-j") == 0) conf.kind = Conf::Juman;
    else if (argValue(argc, argv, i, "--auto-nbest", &v)) {
      // ^(\d+):(\d+):(\d+)$ -> beamSize, autoStep, globalBeam; anything else is ignored like the reference does
      int a = 0, b = 0, c = 0;
      char tail = 0;
      if (std::sscanf(v.c_str(), "%d:%d:%d%c", &a, &b, &c, &tail) == 3 && a >= 0 && b >= 0 && c >= 0) {
        conf.beam = a;
        conf.autoStep = b;
        conf.globalBeam = c;
      }
    } else if (std::strcmp(argv[i], "--partial-input") == 0) conf.partialInput = true;
    else if (std::strcmp(argv[i], "--no-rnn") == 0) conf.noRnn = true;
    else if (argValue(argc, argv, i, "--config", &v) || argValue(argc, argv, i, "-c", &v)) conf.configFile = v;
    else if (argValue(argc, argv, i, "--rnn-model", &v)) conf.rnnModel = v;
    else if (argValue(argc, argv, i, "--rnn-fields", &v)) {
      conf.rnnNames.fields.clear();
      for (size_t p = 0;;) {
        size_t e = v.find(',', p);
        conf.rnnNames.fields.push_back(v.substr(p, e == std::string::npos ? std::string::npos : e - p));
        if (e == std::string::npos) break;
        p = e + 1;
      }
    }
    else if (argValue(argc, argv, i, "--rnn-separator", &v)) conf.rnnNames.separator = v;
    else if (argValue(argc, argv, i, "--rnn-unk", &v)) conf.rnnNames.unkSymbol = v;
    else if (argValue(argc, argv, i, "--rnn-eos", &v)) conf.rnnNames.eosSymbol = v;
    else if (argValue(argc, argv, i, "--rnn-nce-bias", &v)) { conf.rnn.nceBias = std::strtof(v.c_str(), nullptr); conf.rnn.hasNceBias = true; }
    else if (argValue(argc, argv, i, "--rnn-unk-constant", &v)) { conf.rnn.unkConstantTerm = std::strtof(v.c_str(), nullptr); conf.rnn.hasUnkConstantTerm = true; }
    else if (argValue(argc, argv, i, "--rnn-unk-length", &v)) { conf.rnn.unkLengthPenalty = std::strtof(v.c_str(), nullptr); conf.rnn.hasUnkLengthPenalty = true; }
    else if (argValue(argc, argv, i, "--feature-weight-perceptron", &v)) { conf.rnn.perceptronWeight = std::strtof(v.c_str(), nullptr); conf.rnn.hasPerceptronWeight = true; }
    else if (argValue(argc, argv, i, "--feature-weight-rnn", &v)) { conf.rnn.rnnWeight = std::strtof(v.c_str(), nullptr); conf.rnn.hasRnnWeight = true; }
    else if (argValue(argc, argv, i, "--threads", &v)) conf.threads = std::atoi(v.c_str());
    else if (std::strcmp(argv[i], "--no-pipeline") == 0) conf.pipeline = false;
    else if (std::strcmp(argv[i], "--host-format") == 0) conf.hostFormat = true;
    else if (argValue(argc, argv, i, "--pipelines-per-device", &v)) conf.pipelinesPerDevice = std::max(1, std::min(4, std::atoi(v.c_str())));
    else if (std::strcmp(argv[i], "--timing") == 0) conf.timing = true;
    else if (std::strcmp(argv[i], "--no-reserve") == 0) conf.noReserve = true;
    else if (std::strcmp(argv[i], "--no-image-cache") == 0) conf.noImageCache = true;
    else if (std::strcmp(argv[i], "--clean-exit") == 0) conf.cleanExit = true;
    else if (argValue(argc, argv, i, "--output-shards", &v)) conf.outputShards = std::max(1, std::min(4096, std::atoi(v.c_str())));
    else if (argValue(argc, argv, i, "--log-level", &v)) { /* the reference's logging switch: accepted, nothing to log here */ }
    else if (std::strcmp(argv[i], "--help") == 0 || std::strcmp(argv[i], "-h") == 0) conf.help = true;
    else if (argv[i][0] == '-' && argv[i][1] != 0) {
      std::cerr << "unknown option " << argv[i] << "\n";
      return false;
    } else conf.inputs.push_back(argv[i]);
  }
  return true;
}

// JppArgsParser::parseFile (jumanpp_args.cc:183-211): the file's text split at white space
bool readConfigFile(const std::string& path, std::vector<std::string>* tokens) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::string tok;
  while (f >> tok) tokens->push_back(tok);
  return true;
}

bool fileExists(const std::string& p) { return std::ifstream(p, std::ios::binary).good(); }

}  // namespace

int main(int argc, const char** argv) {
  Conf conf;
  std::vector<std::string> cmdline(argv + 1, argv + argc);
  {
    // the config file named on the command line, else ~/.config/jumanpp/jumandic.config, is read first
    Conf probe;
    if (!parseArgList(cmdline, probe)) return 1;
    std::string cfgPath = probe.configFile;
    bool explicitCfg = !cfgPath.empty();
    if (!explicitCfg) {
      if (const char* home = std::getenv("HOME")) cfgPath = std::string(home) + "/.config/jumanpp/jumandic.config";
    }
    std::vector<std::string> tokens;
    if (!cfgPath.empty() && readConfigFile(cfgPath, &tokens)) {
      if (!parseArgList(tokens, conf)) {
        std::cerr << "failed to parse provided config at: " << cfgPath << "\n";
        return 1;
      }
      conf.configFile = cfgPath;
    } else if (explicitCfg) {
      std::cerr << "failed to parse provided config at: " << cfgPath << "\n";
      return 1;
    }
    if (!parseArgList(cmdline, conf)) return 1;
    // -s N asks for N paths: the beam is widened to N if it is narrower (jumanpp_args.cc:253-256)
    if (conf.lattice > 0 && conf.autoStep == 0 && conf.beam < conf.lattice) conf.beam = conf.lattice;
    // fixupModelPath (jumanpp_args.cc:302-338): a relative model path that does not exist is tried next to the config file
    if (!conf.configFile.empty() && !conf.model.empty() && conf.model[0] != '/' && !fileExists(conf.model)) {
      size_t slash = conf.configFile.find_last_of('/');
      std::string rel = (slash == std::string::npos ? std::string(".") : conf.configFile.substr(0, slash)) + "/" + conf.model;
      if (fileExists(rel)) conf.model = rel;
    }
  }
  if (conf.help) {
    std::cerr << "jumanpp_gpu: Juman++ v2 analysis on an MI355X, the batched sibling of jumanpp_v2\n"
                 "  jumanpp_gpu --model=MODEL.jppmdl [options] [INPUT...]   (standard input without INPUT)\n"
                 "General:   -c/--config FILE   -o/--output FILE   --partial-input   --device=N | --devices=0-7\n"
                 "Output:    -j/--juman (default)  -M/--morph  -F/--full-morph  --segment [--segment-separator=S]\n"
                 "           -s/-L/--lattice/--specifics N   --dic-subset   --format=NAME\n"
                 "Analysis:  --beam=5 --global-beam=6 --right-check=1 --right-beam=5 --auto-nbest=BASE:STEP:MAX --no-rnn\n"
                 "RNN:       --rnn-nce-bias=X --rnn-unk-constant=X --rnn-unk-length=X\n"
                 "           --feature-weight-perceptron=X --feature-weight-rnn=X   (0 switches the RNN off)\n"
                 "Batching:  --batch=65536 sentences per GPU launch (lattice output: fewer by default, so that a batch's text stays below ~256 MB), --threads=N format workers, --no-pipeline, --timing, --no-reserve, --no-image-cache, --clean-exit,\n"
                 "           --host-format (text from the host formatters; by default the device prints the top-1 JUMAN format and the -s N lattice format),\n"
                 "           --output-shards=K (files in, file out: the input in K contiguous parts, part k's output in OUT.part000k;\n"
                 "                              one file takes what one writer gives it, however many GPUs feed it),\n"
                 "           --pipelines-per-device=2 (bulk runs: analysis threads, each with its analyzer pair, per GPU)\n";
    return 1;
  }
  if (conf.model.empty()) {
    std::cerr << "Model file was not specified\n";
    return 1;
  }
  const Clock processClock;   // (--timing: where a process' life goes before and after the pipeline, `startup:` / `exit:` lines)
  ModelImage model;
  Status s = model.loadModel(conf.model);
  const double tModelLoaded = processClock.ms();
  if (!s) {
    std::cerr << "failed to load model from disk: " << s << "\n";
    return 1;
  }
  // JumanppEnv::makeAnalyzer: the RNN scorer is used whenever the model carries one (env.cc:86-121)
  AnalyzerConfig acfg;
  acfg.globalBeamSize = conf.globalBeam;
  acfg.rightGbeamCheck = conf.rightCheck;
  acfg.rightGbeamSize = conf.rightBeam;
  if (conf.autoStep > 0) {  // env.setAutoBeam(conf.beamSize, conf.autoStep, conf.globalBeam), jumandic_env.cc:34-36
    acfg.autoBeamBase = conf.beam;
    acfg.autoBeamStep = conf.autoStep;
    acfg.autoBeamMax = conf.globalBeam;
  }
  ScoringConfig sconf;
  sconf.beamSize = conf.beam;
  ScorerDef def;
  RnnScoreWeights weights = model.savedScoreWeights();
  ExternalRnn externalRnn;  // owns the arrays the model points to: lives as long as the analyzers are created from it
  const bool newRnn = !conf.rnnModel.empty() && !conf.noRnn;
  if (newRnn) {  // JumanppExec::init: rnnFactory.make + env.setRnnHolder (jumandic_env.cc:38-48)
    s = externalRnn.load(conf.rnnModel, model, conf.rnnNames);
    if (!s) {
      std::cerr << "failed to load the RNN model: " << s << "\n";
      return 1;
    }
    model.attachExternalRnn(externalRnn.part(), conf.rnn, &weights);
  }
  def.useRnn = model.hasRnn() && !conf.noRnn;
  if (def.useRnn && !newRnn && !conf.rnn.isDefault()) {  // JumanppExec::init: env.setRnnConfig(conf.rnnConfig), jumandic_env.cc:40-42
    bool useRnn = true;
    s = model.applyRnnConfig(conf.rnn, &useRnn, &weights);
    if (!s) {
      std::cerr << "failed to apply the RNN configuration: " << s << "\n";
      return 1;
    }
    def.useRnn = useRnn;
  }
  if (def.useRnn) {
    sconf.numScorers = 2;
    def.scoreWeights = {weights.perceptron, weights.rnn};
  } else if (model.hasRnn() && !conf.noRnn) {
    sconf.numScorers = 1;
    def.scoreWeights = {weights.perceptron};  // RNN switched off by --feature-weight-rnn=0: one scorer, its given weight
  } else {
    sconf.numScorers = 1;
    def.scoreWeights = {1.0f};
  }
  if (conf.threads <= 0) conf.threads = (int)std::min(32u, usableCores());
  // JumanppExec::initOutput (jumandic_env.cc:55-150) and emptyResult (:211-222); one instance per format worker
  StringPiece emptyResult = "# ERROR\nEOS\n";
  const bool latticeFormat = conf.lattice != 0;
  const bool useLattice = latticeFormat || conf.kind == Conf::DicSubset;  // formats that read the whole lattice
  if (!latticeFormat && (conf.kind == Conf::Morph || conf.kind == Conf::FullMorph)) emptyResult = "# ERROR\n";
  if (!latticeFormat && (conf.kind == Conf::Segment || conf.kind == Conf::DicSubset)) emptyResult = "";
  auto makeFormat = [&](Status* st) -> OutputFormat* {
    if (latticeFormat) {
      auto f = new LatticeFormat(conf.lattice == -1 ? conf.beam : conf.lattice);
      *st = f->initialize(&model, def.scoreWeights);
      return f;
    }
    if (conf.kind == Conf::Morph || conf.kind == Conf::FullMorph) {
      auto f = new MorphFormat(conf.kind == Conf::FullMorph);
      *st = f->initialize(&model);
      return f;
    }
    if (conf.kind == Conf::DicSubset) {
      auto f = new SubsetFormat();
      *st = f->initialize(&model);
      return f;
    }
    if (conf.kind == Conf::Segment) {
      auto f = new SegmentedFormat();
      *st = f->initialize(&model, conf.segmentSeparator);
      return f;
    }
    auto f = new JumanFormat();
    *st = f->initialize(&model);
    return f;
  };

  // files in, file out: the sharded pipeline (see the head of this file)
  bool sharded = conf.pipeline && !conf.partialInput && !conf.inputs.empty() && !conf.output.empty() &&
                 conf.output != "-" && std::getenv("JUMANPP_GPU_NO_SHARDED") == nullptr;
  if (sharded) {
    // it maps its inputs and pwrite()s its output: every input must be a regular file, and the output a regular (or
    // new) file that is none of the inputs (O_TRUNC on a mapped input would end in SIGBUS).  FIFOs, /dev/stdin,
    // /dev/stdout, process substitutions and an output that names an input take the general stream pipeline.
    struct stat so;
    const bool haveOut = ::stat(conf.output.c_str(), &so) == 0;
    if (haveOut && !S_ISREG(so.st_mode)) sharded = false;
    for (auto& path : conf.inputs) {
      struct stat si;
      if (::stat(path.c_str(), &si) != 0 || !S_ISREG(si.st_mode)) sharded = false;
      else if (haveOut && si.st_dev == so.st_dev && si.st_ino == so.st_ino) sharded = false;
    }
  }
  if (conf.outputShards > 1 && !sharded) {
    std::cerr << "--output-shards needs regular input files and -o FILE (the file-to-file pipeline)\n";
    return 1;
  }
  std::unique_ptr<std::ofstream> ofile;
  std::ostream* out = &std::cout;
  if (!sharded && !conf.output.empty() && conf.output != "-") {
    ofile.reset(new std::ofstream(conf.output, std::ios::binary));
    out = ofile.get();
  }
  std::ios::sync_with_stdio(false);
  // std::cin is tied to std::cout by default: every read would flush std::cout from the reader thread
  // while the writer thread is inside it
  std::cin.tie(nullptr);

  size_t fileIdx = 0;
  std::unique_ptr<std::ifstream> ifile;
  std::istream* in = &std::cin;
  auto openNext = [&]() -> bool {
    if (fileIdx >= conf.inputs.size()) return false;
    ifile.reset(new std::ifstream(conf.inputs[fileIdx++], std::ios::binary));
    in = ifile.get();
    return true;
  };
  if (!conf.inputs.empty()) openNext();
  // InputOutput::hasNext (jumanpp.cc:82-97)
  auto hasNext = [&]() -> bool {
    for (;;) {
      if (in->good() && in->peek() != std::char_traits<char>::eof()) return true;
      if (conf.inputs.empty() || !openNext()) return false;
    }
  };

  const size_t maxInput = 65535, maxComment = 1024;  // rdr->setMaxSizes(65535, 1024), jumanpp.cc:72
  TrainFieldsIndex tfi;
  PartialExampleReader pexReader;
  if (conf.partialInput) {  // PexStreamReader::initialize(core, '&'), jumanpp.cc:74-77
    s = tfi.initialize(model);
    if (s) s = pexReader.initialize(&tfi, U'&');
    if (!s) {
      std::cerr << "Failed to initialize I/O: " << s << "\n";
      return 1;
    }
  }
  // Four stages, each on its own thread, joined by bounded queues: read a batch | analyse it on the
  // GPU | format it (conf.threads workers, one OutputFormat each) | write it.  Two analyzers alternate
  // so that batch k+1 is analysed while batch k, whose results stay valid until its analyzer's next
  // call, is being formatted.  Output order is the input order.
  const int nAnalyzers = conf.pipeline ? 2 : 1;
  if (conf.devices.empty()) conf.devices.push_back(conf.device);
  // Bulk runs (files in, file out, input of more than eight batches): two pipelines per GPU unless the list was given
  // with repetitions already.  One analysis thread spends 17.9 ms of host time per 65 536-sentence batch inside the
  // library (three waits on the device among them) plus 3 ms of its own around it, for 16 ms of GPU time; a second
  // thread with its own analyzer pair fills those gaps: 3.11 -> 3.69 M sentences/s on one MI355X
  // (profiles/r04_h_cli_stages.txt).  --pipelines-per-device=1 restores one.
  if (sharded && conf.pipelinesPerDevice > 1) {
    size_t inputBytes0 = 0;
    for (auto& path : conf.inputs) {
      struct stat si;
      if (::stat(path.c_str(), &si) == 0) inputBytes0 += (size_t)si.st_size;
    }
    bool repeats = false;
    for (size_t a = 0; a < conf.devices.size(); ++a)
      for (size_t b = a + 1; b < conf.devices.size(); ++b) repeats = repeats || conf.devices[a] == conf.devices[b];
    if (!repeats && inputBytes0 > (size_t)conf.batch * 40 * 8 && inputBytes0 > (size_t{16} << 20)) {
      const std::vector<int> once = conf.devices;
      for (int k = 1; k < conf.pipelinesPerDevice; ++k) conf.devices.insert(conf.devices.end(), once.begin(), once.end());
    }
  }
  const int nDev = (int)conf.devices.size();
  // per device: the second analyzer (a second copy of the model in HBM) is made when a second batch shows
  // up there: a short input pays for one
  std::vector<std::vector<std::unique_ptr<GpuAnalyzer>>> analyzers((size_t)nDev);
  for (auto& v : analyzers) v.resize((size_t)nAnalyzers);
  // The top-1 JUMAN format is printed by the device when the model allows it (format_table.h): every entry row of the
  // dictionary rendered once, here, by the host formatter's own row printer; the analyzers then fetch text instead of
  // node tables and the format workers below have nothing to do.  Lattice / morph / segmented / subset formats,
  // --auto-nbest (several beam groups per batch) and --host-format keep the host formatters.
  JumanFormatTable formatTable;
  bool deviceText = !conf.hostFormat && !latticeFormat && !useLattice && conf.kind == Conf::Juman && acfg.autoBeamStep <= 0 &&
                    std::getenv("JUMANPP_GPU_HOST_FORMAT") == nullptr;
  // What a process derives from the model before its first batch (per-entry T0 records, rendered entry rows) is kept
  // beside the model file by the first run and mapped by the later ones (derived_cache.h)
  DerivedCache derived;
  const bool useImageCache = !conf.noImageCache && std::getenv("JPPGPU_NO_IMAGE_CACHE") == nullptr && !conf.partialInput;
  const bool cacheHit = useImageCache && derived.load(conf.model);
  bool tableFromCache = false;
  if (deviceText && cacheHit && derived.hasFormatTable()) {
    formatTable.adopt(derived.formatTable(), (size_t)derived.formatTableEntries());
    tableFromCache = true;
    if (conf.timing)
      std::cerr << "device_format=1 table_entries=" << formatTable.numEntries() << " rows=" << formatTable.numRows()
                << " blob_bytes=" << formatTable.blobBytes() << " build_ms=0 (image cache)\n";
  }
  if (deviceText && !tableFromCache) {
    Status built = formatTable.build(&model, (unsigned)std::max(1, conf.threads));
    if (!built) deviceText = false;   // (e.g. an UNK maker that rewrites a field the table renders: host formatters)
    if (conf.timing)
      std::cerr << "device_format=" << (deviceText ? 1 : 0) << " table_entries=" << formatTable.numEntries() << " rows=" << formatTable.numRows()
                << " blob_bytes=" << formatTable.blobBytes() << " build_ms=" << formatTable.buildMs() << (built ? "" : " (" + statusText(built) + ")") << "\n";
  }
  // The lattice (-s N) format is printed by the device as well (jppgpu_lattice_table, csrc/k_latfmt.h): the entry-row
  // columns of every dictionary entry rendered once, here; ids, previous ids, ranks and scores come from the kernels.
  // --auto-nbest (N differs per sentence), N > 64, --global-beam=0 (no score cells) and --host-format keep the host
  // formatter, which reads the N best paths gathered on the device (jppgpu_result_fetch_nbest).
  const int32_t latticeN = conf.lattice == -1 ? conf.beam : conf.lattice;
  LatticeFormatTable latticeTable;
  bool deviceLattice = !conf.hostFormat && latticeFormat && acfg.autoBeamStep <= 0 && latticeN >= 1 && latticeN <= 64 &&
                       acfg.globalBeamSize > 0 && !conf.partialInput && std::getenv("JUMANPP_GPU_HOST_FORMAT") == nullptr;
  bool latticeFromCache = false;
  if (deviceLattice && cacheHit && derived.hasLatticeTable() &&
      latticeTable.adopt(derived.latticeTable(), (size_t)derived.latticeTableEntries(), def.scoreWeights)) {
    latticeFromCache = true;
    if (conf.timing)
      std::cerr << "device_lattice_format=1 table_entries=" << latticeTable.numEntries() << " rows=" << latticeTable.numRows()
                << " blob_bytes=" << latticeTable.blobBytes() << " build_ms=0 (image cache)\n";
  }
  if (deviceLattice && !latticeFromCache) {
    Status built = latticeTable.build(&model, def.scoreWeights, (unsigned)std::max(1, conf.threads));
    if (!built) deviceLattice = false;
    if (conf.timing)
      std::cerr << "device_lattice_format=" << (deviceLattice ? 1 : 0) << " table_entries=" << latticeTable.numEntries() << " rows=" << latticeTable.numRows()
                << " blob_bytes=" << latticeTable.blobBytes() << " build_ms=" << latticeTable.buildMs() << (built ? "" : " (" + statusText(built) + ")") << "\n";
  }
  // text bytes per input byte, for the plans below: a morpheme covers ~5.5 input bytes; the lattice prints a line of
  // row text + ~70 bytes of ids and scores for every node on one of the N paths (3-4 times the top-1 path at N = 32)
  const double latticeTextPerByte =
      !deviceLattice || latticeTable.numRows() == 0
          ? 0.0
          : 1.2 * ((double)latticeTable.blobBytes() / (double)latticeTable.numRows() + 70.0) / 5.5 * (1.0 + 0.075 * std::min<int32_t>(latticeN, 32));
  // (only a regular file is sampled: a FIFO, /dev/stdin or a process substitution would lose the sampled megabyte, or
  // block on the second open once its writer is gone -- those keep the default batch, as the reference reads any stream)
  struct stat sampleStat;
  const bool sampleable = !conf.inputs.empty() && ::stat(conf.inputs[0].c_str(), &sampleStat) == 0 && S_ISREG(sampleStat.st_mode);
  if (latticeFormat && !conf.batchGiven && sampleable) {
    // The lattice format reads the N best paths of every sentence as the device gathers them (jppgpu_result_fetch_nbest:
    // 64 B per path and node).  At beam 32 on 220-codepoint sentences that is 0.2 MB per sentence; a 16 384-sentence
    // batch moves 3.3 GB through fresh host pages and the analysis stage spends five times the GPU's time on it
    // (tools/gpu_cli_lattice_probe.py: 20.8 k sentences/s at --batch=16384, 36.2 k at 8192, 48.8 k at 4096).  Unless the
    // user names a batch size, a batch's gathered paths stay below ~1 GB: nodes per path estimated from the mean line.
    std::ifstream f(conf.inputs[0], std::ios::binary);
    std::vector<char> buf(size_t{1} << 20);
    f.read(buf.data(), (std::streamsize)buf.size());
    const size_t got = (size_t)f.gcount();
    size_t lines = 0;
    for (size_t i = 0; i < got; ++i) lines += buf[i] == '\n';
    if (got != 0 && lines != 0) {
      const double meanLine = (double)got / (double)lines;
      const double nodesPerPath = std::max(4.0, meanLine / 5.0);   // ~3 bytes per codepoint, ~1.7 codepoints per node
      const double paths = (double)latticeN;
      // (device text: what crosses PCIe is the text itself; a batch's text stays below ~256 MB, the size of one of the
      // page-locked blocks it is copied into)
      const double perSentence = deviceLattice ? meanLine * latticeTextPerByte * 4.0 : std::max(1.0, std::min(paths, 64.0)) * nodesPerPath * 64.0;
      size_t cap = (size_t)(1.0e9 / perSentence);
      cap = std::max<size_t>(1024, cap / 1024 * 1024);
      if (cap < conf.batch) conf.batch = cap;
    }
  }
  // The batches of a file-to-file run, sized before anything is allocated (the inputs are regular files): lines per batch
  // x the mean line of a sample, with a margin; the JUMAN text of a batch from the mean row of the format table (a
  // morpheme covers ~5.5 input bytes and prints about one row).  Used twice: the page-locked text blocks are pinned on
  // a thread of their own from HERE on -- beside the analyzers being made, which is the slower half of the start-up --
  // and every analyzer reserves its device buffers before the pipeline's clock starts (below).
  uint64_t planBatchBytes = 0;
  uint32_t planBatchLines = 0;
  float planTextPerByte = 0.f;
  struct Joiner {
    std::thread t;
    ~Joiner() {
      if (t.joinable()) t.join();
    }
  } prepin;
  if (sharded && !conf.noReserve) {
    size_t inputTotal = 0, sampleBytes = 0, sampleLines = 0;
    for (auto& path : conf.inputs) {
      struct stat si;
      if (::stat(path.c_str(), &si) == 0) inputTotal += (size_t)si.st_size;
    }
    {
      std::ifstream f(conf.inputs[0], std::ios::binary);
      std::vector<char> buf(size_t{4} << 20);
      f.read(buf.data(), (std::streamsize)buf.size());
      sampleBytes = (size_t)f.gcount();
      for (size_t i = 0; i < sampleBytes; ++i) sampleLines += buf[i] == '\n';
      if (sampleBytes && (sampleLines == 0 || buf[sampleBytes - 1] != '\n')) ++sampleLines;
    }
    const double meanLine = sampleLines ? (double)sampleBytes / (double)sampleLines : 64.0;
    planBatchBytes = std::min<uint64_t>((uint64_t)inputTotal + 64, (uint64_t)((double)conf.batch * meanLine * 1.15) + 65536);
    planBatchLines = (uint32_t)std::min<uint64_t>(conf.batch, (uint64_t)((double)inputTotal / std::max(1.0, meanLine - 1.0)) + 16);
    if ((deviceText && formatTable.numRows() > 0) || deviceLattice) {
      planTextPerByte = deviceLattice ? (float)latticeTextPerByte : (float)(1.1 * ((double)formatTable.blobBytes() / (double)formatTable.numRows()) / 5.5);
      const uint64_t textBytes = (uint64_t)((double)planBatchBytes * planTextPerByte) + 64 * (uint64_t)planBatchLines + 4096;
      const size_t nBatches = (size_t)((double)inputTotal / std::max(1.0, (double)planBatchBytes / 1.15)) + 1;
      // per pipeline: the block being filled, one queued, one being written
      const uint32_t blocks = (uint32_t)std::min<size_t>(nBatches, 3 * conf.devices.size());
      const int dev0 = conf.devices.empty() ? conf.device : conf.devices[0];
      prepin.t = std::thread([=]() { (void)jppgpu_host_prepin(dev0, textBytes, blocks); });
    }
  }
  auto makeAnalyzer = [&](int d, int a) -> Status {
    analyzers[d][a].reset(new GpuAnalyzer());
    // the lattice format reads the N best paths only: they are gathered on the device (N = what it prints)
    if (latticeFormat && !deviceLattice) analyzers[d][a]->setLatticeNBest(latticeN);
    // one copy of the model per GPU: a later analyzer of the same physical device uses the first one's
    const GpuAnalyzer* donor = nullptr;
    for (int d2 = 0; d2 < nDev && donor == nullptr; ++d2)
      if (conf.devices[d2] == conf.devices[d] && analyzers[d2][0] && analyzers[d2][0].get() != analyzers[d][a].get() &&
          analyzers[d2][0]->ready())
        donor = analyzers[d2][0].get();
    if (donor == nullptr && cacheHit && derived.memo() != nullptr)
      analyzers[d][a]->setT0MemoImage(derived.memo(), derived.memoBytes(), derived.memoSlots());
    else if (donor == nullptr && useImageCache && d == 0 && a == 0)
      analyzers[d][a]->setKeepT0MemoImage(true);   // (no records in a cache: this run writes them, below)
    Status made = analyzers[d][a]->initialize(&model, acfg, sconf, &def, conf.devices[d], donor);
    if (made && deviceText) {
      if (donor == nullptr) made = analyzers[d][a]->setFormatTable(formatTable.view());
      if (made) {
        analyzers[d][a]->setTextMode(true);
        analyzers[d][a]->setDeferredText(sharded);
      }
    }
    if (made && deviceLattice) {
      if (donor == nullptr) made = analyzers[d][a]->setLatticeTable(latticeTable.view());
      if (made && !analyzers[d][a]->setLatticeTextMode(latticeN)) made = Status::InvalidState("lattice text mode was refused");
      if (made) analyzers[d][a]->setDeferredText(sharded);
    }
    return made;
  };
  for (int d = 0; d < nDev; ++d) {
    s = makeAnalyzer(d, 0);
    if (!s) {
      std::cerr << "failed to initialize the analyzer on device " << conf.devices[d] << ": " << s << "\n";
      return 1;
    }
  }
  if (conf.timing && useImageCache) std::cerr << std::string(cacheHit ? "image_cache=hit\n" : "image_cache=miss\n");
  if (conf.timing) {
    std::ostringstream ln;
    ln << "startup: model_map_ms=" << tModelLoaded << " first_analyzers_ready_ms=" << processClock.ms() << "\n";
    std::cerr << ln.str();
  }
  // The cache is written by the run that misses it -- and REwritten by a run that hit a cache lacking a part it needs
  // (the first run of a model may have had no device text: -s N, --host-format, another output format; or no memo): the
  // missing part is added to what the mapping holds, so that the next start finds both.  The records come from the
  // library's own host copy (valid until the last context goes; this process never calls jppgpu_ctx_set_weights, which
  // would refill them in place), the mapped parts from `derived`, which outlives the writer.
  Joiner cacheWriter;
  if (useImageCache && analyzers[0][0]) {
    const void* memo = cacheHit ? derived.memo() : nullptr;
    uint64_t memoBytes = cacheHit ? derived.memoBytes() : 0;
    uint32_t memoSlots = cacheHit ? derived.memoSlots() : 0;
    bool newPart = !cacheHit;
    if (memo == nullptr) {
      const bool haveMemo = analyzers[0][0]->exportT0MemoImage(&memo, &memoBytes, &memoSlots);
      if (!haveMemo) memo = nullptr;
      newPart = newPart || haveMemo;
    }
    const jppgpu_format_table* tbl = nullptr;
    uint64_t entries = 0;
    if (deviceText) {
      tbl = &formatTable.view();
      entries = formatTable.numEntries();
      newPart = newPart || !tableFromCache;
    } else if (cacheHit && derived.hasFormatTable()) {
      tbl = &derived.formatTable();
      entries = derived.formatTableEntries();
    }
    const jppgpu_lattice_table* lat = nullptr;
    uint64_t latEntries = 0;
    if (deviceLattice) {
      lat = &latticeTable.view();
      latEntries = latticeTable.numEntries();
      newPart = newPart || !latticeFromCache;
    } else if (cacheHit && derived.hasLatticeTable()) {
      lat = &derived.latticeTable();
      latEntries = derived.latticeTableEntries();
    }
    if (newPart && (memo != nullptr || tbl != nullptr || lat != nullptr)) {
      const std::string modelPath = conf.model;
      const bool timing = conf.timing;
      cacheWriter.t = std::thread([=]() {
        const bool ok = DerivedCache::store(modelPath, memo, memoBytes, memoSlots, tbl, entries, lat, latEntries);
        if (timing) {
          // (write(2), not std::cerr: the main thread prints through the unsynchronised stream object at the same time,
          // and once in a few hundred runs this line was lost)
          const std::string line = std::string("image cache ") + (ok ? "written" : "not written") + " for " + modelPath + "\n";
          const ssize_t w = ::write(2, line.data(), line.size());
          (void)w;
        }
      });
    }
  }
  std::vector<std::unique_ptr<OutputFormat>> formats;
  for (int t = 0; t < conf.threads; ++t) {
    formats.emplace_back(makeFormat(&s));
    if (!s) {
      std::cerr << "Failed to initialize I/O: " << s << "\n";
      return 1;
    }
  }

  if (sharded) {
    // The second analyzer of every device (a second context: its own workspaces and, today, its own copy of the model
    // tables) is made here, all of them in parallel, when the input is large enough to need it -- before the pipeline's
    // clock starts: made while the first batches ran it stalled the second batch of a device for the 0.3 s a context
    // of a 1 M-entry model takes (profiles/r04_f_cli_stages.txt: 16 batches in 611 ms at 16 ms per batch).
    {
      size_t inputBytes0 = 0;
      for (auto& path : conf.inputs) {
        struct stat si;
        if (::stat(path.c_str(), &si) == 0) inputBytes0 += (size_t)si.st_size;
      }
      if (nAnalyzers > 1 && inputBytes0 > (size_t)conf.batch * 16 * (size_t)nDev) {
        std::vector<std::future<Status>> made;
        for (int d = 0; d < nDev; ++d) made.emplace_back(std::async(std::launch::async, [&, d]() { return makeAnalyzer(d, 1); }));
        for (int d = 0; d < nDev; ++d)
          if (!made[(size_t)d].get()) analyzers[(size_t)d][1].reset();   // (no HBM for it: the device thread carries on with one)
      }
    }
    Clock clock;
    // every device's format workers have OutputFormat objects of their own
    while ((int)formats.size() < nDev * std::max(1, conf.threads / nDev)) {
      formats.emplace_back(makeFormat(&s));
      if (!s) {
        std::cerr << "Failed to initialize I/O: " << s << "\n";
        return 1;
      }
    }
    std::vector<std::unique_ptr<MappedFile>> maps;
    for (auto& path : conf.inputs) {
      maps.emplace_back(new MappedFile());
      if (!maps.back()->open(path)) {
        std::cerr << "could not map the input file " << path << "\n";
        return 1;
      }
    }
    // Every analyzer takes the buffers of its batches at their final size NOW, all of them in parallel and before the
    // clock of the pipeline starts (GpuAnalyzer::reserve): the input is mapped, so the bytes of a batch are known ahead --
    // lines per batch x the mean line of a sample, with a margin -- and the text of a batch from the mean row of the
    // format table.  Up to round 4 the buffers grew inside the first batches (hipMalloc / hipHostMalloc stalls of
    // 0.2-0.9 s in batches 0-2, profiles/r04_t_cli_stages.txt) and every batch waited three times for the totals that
    // size them; now a batch is one enqueue and allocates nothing.  (--no-reserve: the round-4 behaviour.)
    // The page-locked text blocks are pinned on a thread of their own since process start; normally they are ready long
    // before this point (`prepin: waited_ms=0.00x`).  If a box pins slowly, the wait belongs to the start-up like the
    // reservation below, not to the first batches (page-locking and kernel launches share locks of the runtime).
    if (prepin.t.joinable()) {
      const double p0 = clock.ms();
      prepin.t.join();
      if (conf.timing) {
        std::ostringstream ln;
        ln << "prepin: waited_ms=" << clock.ms() - p0 << "\n";
        std::cerr << ln.str();
      }
      clock = Clock();
    }
    if (!conf.noReserve) {
      const uint64_t batchBytes = planBatchBytes;
      const uint32_t batchLines = planBatchLines;
      const float textPerByte = planTextPerByte;
      const double r0 = clock.ms();
      std::vector<std::future<Status>> reserved;
      for (int d = 0; d < nDev; ++d)
        for (int a = 0; a < nAnalyzers; ++a)
          if (analyzers[(size_t)d][(size_t)a])
            reserved.emplace_back(std::async(std::launch::async, [&, d, a]() {
              return analyzers[(size_t)d][(size_t)a]->reserve(batchLines, batchBytes, textPerByte, 0);
            }));
      for (auto& f : reserved) {
        Status rs = f.get();
        if (!rs && conf.timing) std::cerr << "reserve failed (the batches size their buffers themselves): " << rs << "\n";
      }
      if (conf.timing)
        std::cerr << "reserve: batch_lines=" << batchLines << " batch_bytes=" << batchBytes << " text_per_byte=" << textPerByte
                  << " analyzers=" << reserved.size() << " ms=" << clock.ms() - r0 << "\n";
      clock = Clock();   // (like the model load and the analyzers themselves: not the pipeline's time)
    }
    // One output file -- or, with --output-shards=K, K of them: the input is cut into K contiguous parts of about equal
    // size at example boundaries, part k is analysed into OUT.part000k, and `cat OUT.part*` is the one-file output.
    // Buffered writes to ONE file are serialised by the kernel on the file's inode whatever the number of writer threads
    // (tools/host_write_ceiling.py on the MI355X box: 14 GB/s with 1 .. 16 threads, i.e. 5.9 M sentences/s of JUMAN text,
    // the rate of 1.4 GPUs; 28 / 55 / 98 / 160 GB/s into 2 / 4 / 8 / 16 files): a node of eight GPUs needs the shards.
    const int nShards = conf.outputShards;
    std::vector<int> ofds;
    for (int k = 0; k < nShards; ++k) {
      char suffix[32];
      std::snprintf(suffix, sizeof(suffix), ".part%04d", k);
      const std::string path = nShards == 1 ? conf.output : conf.output + suffix;
      const int fd = ::open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
      if (fd < 0) {
        std::cerr << "could not open the output file " << path << "\n";
        return 1;
      }
      ofds.push_back(fd);
    }
    // the parts: (file, begin, end) runs of whole examples.  A cut falls on the line start at or behind its byte offset,
    // moved back over the "# " comment lines in front of it (they belong to the example that follows them).
    struct PartRun {
      size_t file;
      const char* begin;
      const char* end;
    };
    std::vector<std::vector<PartRun>> parts((size_t)nShards);
    if (!maps.empty()) {
      size_t total = 0;
      for (auto& mf : maps) total += mf->size;
      // bounds[k] .. bounds[k + 1]: part k, as (file, position) pairs in input order
      std::vector<std::pair<size_t, const char*>> bounds;
      bounds.emplace_back(0, maps[0]->data);
      for (int k = 1; k < nShards; ++k) {
        size_t want = (size_t)((double)total * k / nShards), f = 0;
        while (f + 1 < maps.size() && want >= maps[f]->size) want -= maps[f++]->size;
        const char* const b = maps[f]->data;
        const char* const e = b + maps[f]->size;
        const char* p = b + std::min(want, maps[f]->size);
        if (p > b && p < e) {   // to the next line start
          const char* nl = static_cast<const char*>(memchr(p - 1, '\n', (size_t)(e - (p - 1))));
          p = nl ? nl + 1 : e;
        }
        while (p > b) {   // back over the comment lines in front of it
          const char* prevEnd = p - 1;   // the newline that ends the previous line (or the last byte of a file without one)
          if (*prevEnd != '\n') break;
          const char* ls = prevEnd;
          while (ls > b && ls[-1] != '\n') --ls;
          if (prevEnd - ls > 2 && ls[0] == '#' && ls[1] == ' ') p = ls;
          else break;
        }
        if (f < bounds.back().first || (f == bounds.back().first && p < bounds.back().second)) bounds.push_back(bounds.back());
        else bounds.emplace_back(f, p);
      }
      bounds.emplace_back(maps.size() - 1, maps.back()->data + maps.back()->size);
      for (int k = 0; k < nShards; ++k) {
        const auto& lo = bounds[(size_t)k];
        const auto& hi = bounds[(size_t)k + 1];
        for (size_t f = lo.first; f <= hi.first; ++f) {
          const char* begin = f == lo.first ? lo.second : maps[f]->data;
          const char* end = f == hi.first ? hi.second : maps[f]->data + maps[f]->size;
          if (begin < end) parts[(size_t)k].push_back(PartRun{f, begin, end});
        }
      }
    }
    std::vector<std::unique_ptr<BoundedQueue<std::unique_ptr<ShardJob>>>> readQ, fmtQ, writeQ;
    std::vector<std::unique_ptr<Semaphore>> freeAnalyzers;
    for (int d = 0; d < nDev; ++d) {
      readQ.emplace_back(new BoundedQueue<std::unique_ptr<ShardJob>>(2));
      fmtQ.emplace_back(new BoundedQueue<std::unique_ptr<ShardJob>>(1));
      writeQ.emplace_back(new BoundedQueue<std::unique_ptr<ShardJob>>(2));
      freeAnalyzers.emplace_back(new Semaphore(nAnalyzers));
    }
    std::atomic<long long> scanUs(0), prepUs(0), formatUs(0), writeUs(0), gpuUs(0), fmtCountUs(0), fmtWriteUs(0);
    std::vector<double> analyzeMsDev((size_t)nDev, 0.0);
    auto us = [&]() { return (long long)(clock.ms() * 1000.0); };
    // The second analyzer of a device (a second copy of the model and its per-entry tables in HBM: ~0.15 s) is made on
    // a thread of its own while the first batch is read and analysed, when the input is large enough to need it; made
    // lazily on the analysis thread it cost a quarter of a 1 M-line run (profiles/r03_v_cli_batches.txt).
    size_t inputBytes = 0;
    for (auto& mf : maps) inputBytes += mf->size;
    std::vector<std::future<Status>> secondAnalyzer((size_t)nDev);   // (made above; a lazily made one is the fallback)
    (void)inputBytes;

    // scanner: batches of conf.batch examples (an example = its "# " comment lines + one other line,
    // PlainStreamReader::readExample, stream_reader.cc:12-38), dealt to the devices in turn
    std::thread scanner([&]() {
      // one cursor per part; the parts take turns, so that every output file has batches under way all the time
      struct Cursor {
        size_t run = 0;
        const char* p = nullptr;
        size_t seq = 0;
      };
      std::vector<Cursor> cur((size_t)nShards);
      for (int k = 0; k < nShards; ++k)
        if (!parts[(size_t)k].empty()) cur[(size_t)k].p = parts[(size_t)k][0].begin;
      size_t dealt = 0;
      for (bool any = true; any;) {
        any = false;
        for (int k = 0; k < nShards; ++k) {
          Cursor& c = cur[(size_t)k];
          const std::vector<PartRun>& runs = parts[(size_t)k];
          while (c.run < runs.size() && c.p >= runs[c.run].end) {
            ++c.run;
            if (c.run < runs.size()) c.p = runs[c.run].begin;
          }
          if (c.run >= runs.size()) continue;
          any = true;
          const char* p = c.p;
          const char* const end = runs[c.run].end;
          // (the end of a run that is the end of its FILE: comment lines there make an example with an empty input)
          const bool fileEnd = end == maps[runs[c.run].file]->data + maps[runs[c.run].file]->size;
          const long long t0 = us();
          const char* q = p;
          size_t examples = 0;
          std::unique_ptr<ShardJob> job(new ShardJob());
          job->inputs.reserve(conf.batch);
          job->comments.reserve(conf.batch);
          StringPiece comment("", 0);
          // one pass over the lines: the batch is cut here AND split into its examples (the device threads used to walk
          // the same bytes a second time, 1.1 ms per batch on the thread that feeds the GPU)
          while (q < end && examples < conf.batch) {
            const char* nl = static_cast<const char*>(memchr(q, '\n', (size_t)(end - q)));
            const char* lineEnd = nl ? nl : end;
            const StringPiece line(q, (size_t)(lineEnd - q));
            q = nl ? nl + 1 : end;
            if (line.size() > 2 && line[0] == '#' && line[1] == ' ') {
              comment = line;
              if (q < end || !fileEnd) continue;
              job->inputs.push_back(StringPiece("", 0));   // comment lines at the end of a file: an example with an empty input
            } else {
              job->inputs.push_back(line);
            }
            job->comments.push_back(comment);
            const size_t i = job->inputs.size() - 1;
            if (comment.size() > maxComment)
              job->readErrors.emplace_back(i, Status::InvalidParameter() << "Comment size was: " << comment.size() << " which is more than max: " << maxComment);
            else if (job->inputs[i].size() > maxInput)
              job->readErrors.emplace_back(i, Status::InvalidParameter() << "Input size was: " << job->inputs[i].size() << " which is more than max: " << maxInput);
            comment = StringPiece("", 0);
            ++examples;
          }
          c.p = q;
          if (job->inputs.empty()) continue;   // (a run of nothing but comment lines in front of a cut cannot happen: cuts move back over them)
          job->lastReadOk = job->readErrors.empty() || job->readErrors.back().first + 1 != job->inputs.size();
          job->seq = c.seq++;
          job->shard = k;
          job->data = p;
          job->bytes = (size_t)(q - p);
          job->device = (int)(dealt % (size_t)nDev);
          ++dealt;
          scanUs += us() - t0;
          readQ[job->device]->push(std::move(job));
        }
      }
      for (auto& qd : readQ) qd->close();
    });

    // per device: split the batch into examples, analyse
    std::vector<std::thread> gpus;
    for (int d = 0; d < nDev; ++d) {
      gpus.emplace_back([&, d]() {
        std::unique_ptr<ShardJob> job;
        int next = 0, live = nAnalyzers;
        while (readQ[d]->pop(&job)) {
          long long t0 = us();
          prepUs += us() - t0;
          freeAnalyzers[d]->acquire();
          job->analyzer = next;
          const double a0 = clock.ms();
          const bool pendingSecond = job->analyzer == 1 && secondAnalyzer[(size_t)d].valid();
          if (pendingSecond || !analyzers[d][job->analyzer]) {
            Status made = pendingSecond ? secondAnalyzer[(size_t)d].get() : makeAnalyzer(d, job->analyzer);
            if (!made) {   // (no HBM for a second copy of the model) carry on with the first analyzer alone
              analyzers[d][job->analyzer].reset();
              freeAnalyzers[d]->acquire();
              live = 1;
              job->analyzer = 0;
            }
          }
          next = (job->analyzer + 1) % live;
          {
            std::vector<StringPiece> pieces(job->inputs);
            for (auto& e : job->readErrors) pieces[e.first] = StringPiece("", 0);
            GpuAnalyzer& analyzer = *analyzers[d][job->analyzer];
            job->batchStatus = analyzer.analyzeBatch(pieces, useLattice);
            float ms[8];
            analyzer.lastTimings(ms);
            job->gpuMs = ms[7];
          }
          analyzeMsDev[d] += clock.ms() - a0;
          fmtQ[d]->push(std::move(job));
        }
        fmtQ[d]->close();
      });
    }

    // sequencer state: batches take their output offset in input order
    std::mutex seqMu;
    std::condition_variable seqCv;
    std::vector<size_t> seqNext((size_t)nShards, 0);   // per output shard
    std::vector<uint64_t> outTotal((size_t)nShards, 0);
    std::vector<int> shardResult((size_t)nShards, -1);
    size_t sentences = 0;
    int result = 0;

    // per device: format with the device's own workers, take the offset, hand the batch to the device's writer
    const int perDev = std::max(1, conf.threads / nDev);
    std::vector<std::thread> formatters;
    for (int d = 0; d < nDev; ++d) {
      formatters.emplace_back([&, d]() {
        WorkerPool pool(perDev);
        std::unique_ptr<ShardJob> job;
        const size_t kChunk = 64;
        while (fmtQ[d]->pop(&job)) {
          const long long t0 = us();
          const size_t n = job->inputs.size();
          if ((deviceText || deviceLattice) && job->batchStatus.isOk()) {
            // the device formats: this thread runs the format kernels and copies the text (while the analysis thread is
            // on the device's other analyzer), then cuts the batch's bytes into the segments the writer puts out --
            // one run per stretch of sentences without a comment line, a failed read or an error message
            GpuAnalyzer& an = *analyzers[d][job->analyzer];
            Status fs = an.fetchText();
            {
              float fm[2];
              an.lastFormatTimings(fm);
              fmtCountUs += (long long)(fm[0] * 1000.0);
              fmtWriteUs += (long long)(fm[1] * 1000.0);
            }
            job->errors.assign(1, std::string());
            std::string& errors = job->errors[0];
            job->outBytes = 0;
            if (!fs) {
              job->batchStatus = fs;
            } else {
              const jppgpu_text_view& tv = an.batchText();
              const uint32_t* heads = an.batchTextHeads();
              auto seg = [&](const char* p, size_t len) {
                if (len == 0) return;
                if (!job->segments.empty()) {
                  struct iovec& last = job->segments.back();
                  if (static_cast<const char*>(last.iov_base) + last.iov_len == p) {
                    last.iov_len += len;
                    job->outBytes += len;
                    return;
                  }
                }
                struct iovec v;
                v.iov_base = const_cast<char*>(p);
                v.iov_len = len;
                job->segments.push_back(v);
                job->outBytes += len;
              };
              auto bad = job->readErrors.begin();
              for (size_t i = 0; i < n; ++i) {
                if (bad != job->readErrors.end() && bad->first == i) {
                  errors += "failed to read an example: " + statusText(bad->second);
                  ++bad;
                  continue;   // (analysed as an empty line; nothing of it is printed)
                }
                if (tv.status[i] != JPPGPU_SENT_OK) errors += statusText(an.sentenceStatus(i));   // (its text is the error result)
                const StringPiece& cm = job->comments[i];
                // lattice text: a comment takes the place of the "# MA-SCORE" line the text starts with, and a sentence
                // without that line (empty input: "EOS" alone) prints none (lattice_format.cc:87-120)
                const size_t head = heads != nullptr ? heads[i] : 0;
                size_t skip = 0;
                if (cm.size() >= 2 && tv.status[i] == JPPGPU_SENT_OK && (heads == nullptr || head != 0)) {
                  seg(cm.data(), cm.size());   // "# comment", as it stands in the mapped input
                  seg("\n", 1);
                  skip = head;
                }
                seg(tv.text + tv.offsets[i] + skip, (size_t)(tv.offsets[i + 1] - tv.offsets[i]) - skip);
              }
              job->deviceText = an.takeText();
            }
          }
          if ((deviceText || deviceLattice) && job->batchStatus.isOk()) {
            freeAnalyzers[d]->release();
            formatUs += us() - t0;
            std::unique_lock<std::mutex> l(seqMu);
            seqCv.wait(l, [&] { return seqNext[(size_t)job->shard] == job->seq; });
            job->outOffset = outTotal[(size_t)job->shard];
            outTotal[(size_t)job->shard] += job->outBytes;
            sentences += n;
            gpuUs += (long long)(job->gpuMs * 1000.0);
            shardResult[(size_t)job->shard] = job->lastReadOk ? 0 : 1;
            for (auto& e : job->errors)
              if (!e.empty()) std::cerr << e;
            ++seqNext[(size_t)job->shard];
            seqCv.notify_all();
            l.unlock();
            writeQ[d]->push(std::move(job));
            continue;
          }
          const GpuAnalyzer& analyzer = *analyzers[d][job->analyzer];
          const size_t nChunks = (n + kChunk - 1) / kChunk;
          job->text.assign(nChunks, std::string());
          job->errors.assign(nChunks, std::string());
          std::atomic<size_t> nextChunk(0);
          ShardJob* jp = job.get();
          pool.run([&, jp](int t) {
            OutputFormat* format = formats[(size_t)(d * perDev + t) % formats.size()].get();
            for (size_t c; (c = nextChunk.fetch_add(1)) < nChunks;) {
              std::string& text = jp->text[c];
              std::string& errors = jp->errors[c];
              auto bad = jp->readErrors.begin();
              while (bad != jp->readErrors.end() && bad->first < c * kChunk) ++bad;
              for (size_t i = c * kChunk; i < std::min(n, (c + 1) * kChunk); ++i) {
                if (bad != jp->readErrors.end() && bad->first == i) {
                  errors += "failed to read an example: " + statusText(bad->second);
                  ++bad;
                  continue;
                }
                Status st = jp->batchStatus.isOk() ? analyzer.sentenceStatus(i) : jp->batchStatus;
                if (!st) {
                  errors += statusText(st);
                  text.append(emptyResult.data(), emptyResult.size());
                  continue;
                }
                const StringPiece& cm = jp->comments[i];
                const StringPiece comment = cm.size() < 2 ? StringPiece("") : StringPiece(cm.data() + 2, cm.size() - 2);
                st = format->format(analyzer, i, comment);
                if (!st) errors += statusText(st);
                else {
                  StringPiece r = format->result();
                  text.append(r.data(), r.size());
                }
              }
            }
          });
          job->outBytes = 0;
          for (auto& t : job->text) job->outBytes += t.size();
          freeAnalyzers[d]->release();   // the analyzer's results are no longer needed
          formatUs += us() - t0;
          {
            std::unique_lock<std::mutex> l(seqMu);
            seqCv.wait(l, [&] { return seqNext[(size_t)job->shard] == job->seq; });
            job->outOffset = outTotal[(size_t)job->shard];
            outTotal[(size_t)job->shard] += job->outBytes;
            sentences += n;
            gpuUs += (long long)(job->gpuMs * 1000.0);
            shardResult[(size_t)job->shard] = job->lastReadOk ? 0 : 1;   // the reference's exit code is that of the last example read
            for (auto& e : job->errors)
              if (!e.empty()) std::cerr << e;   // (in input order; with several output shards: in each shard's order)
            ++seqNext[(size_t)job->shard];
            seqCv.notify_all();
          }
          writeQ[d]->push(std::move(job));
        }
        writeQ[d]->close();
      });
    }

    std::atomic<bool> writeFailed(false);
    std::vector<std::thread> writers;
    // A batch is 150 MB of text: one thread copies that into the page cache in ~20 ms -- longer than the GPU needs for the
    // batch (profiles/r04_g_cli_stages.txt: the writer was the slowest stage at 3.1 M sentences/s).  The writer of a device
    // hands out byte ranges of the batch to a few helpers; every range is written at its own file offset.
    const int writeHelpers = std::max(1, std::min(4, conf.threads / std::max(1, nDev)));
    for (int d = 0; d < nDev; ++d) {
      writers.emplace_back([&, d]() {
        WorkerPool pool(writeHelpers);
        std::unique_ptr<ShardJob> job;
        while (writeQ[d]->pop(&job)) {
          const long long t0 = us();
          // the host formatters' chunks are segments like the device text's
          if (job->segments.empty())
            for (auto& t : job->text)
              if (!t.empty()) {
                struct iovec v;
                v.iov_base = const_cast<char*>(t.data());
                v.iov_len = t.size();
                job->segments.push_back(v);
              }
          const std::vector<struct iovec>& segs = job->segments;
          std::vector<uint64_t> start(segs.size() + 1, 0);
          for (size_t k = 0; k < segs.size(); ++k) start[k + 1] = start[k] + segs[k].iov_len;
          const uint64_t total = start[segs.size()];
          const uint64_t base = job->outOffset;
          pool.run([&](int t) {
            const uint64_t lo = total * (uint64_t)t / (uint64_t)writeHelpers, hi = total * (uint64_t)(t + 1) / (uint64_t)writeHelpers;
            if (lo >= hi) return;
            size_t k = (size_t)(std::upper_bound(start.begin(), start.end(), lo) - start.begin()) - 1;
            uint64_t pos = lo;
            while (pos < hi && !writeFailed) {
              // pwritev takes at most IOV_MAX segments and may write less than it was given
              struct iovec part[256];
              int cnt = 0;
              uint64_t p2 = pos;
              size_t k2 = k;
              while (cnt < 256 && p2 < hi && k2 < segs.size()) {
                const uint64_t inSeg = p2 - start[k2];
                const uint64_t len = std::min<uint64_t>(segs[k2].iov_len - inSeg, hi - p2);
                part[cnt].iov_base = static_cast<char*>(segs[k2].iov_base) + inSeg;
                part[cnt].iov_len = (size_t)len;
                ++cnt;
                p2 += len;
                ++k2;
              }
              const ssize_t w = pwritev(ofds[(size_t)job->shard], part, cnt, (off_t)(base + pos));
              if (w <= 0) {
                writeFailed = true;
                return;
              }
              pos += (uint64_t)w;
              while (k < segs.size() && start[k + 1] <= pos) ++k;
            }
          });
          job->deviceText.reset();   // the text block goes back to its analyzer's pool
          writeUs += us() - t0;
        }
      });
    }
    scanner.join();
    for (auto& t : gpus) t.join();
    for (auto& t : formatters) t.join();
    for (auto& t : writers) t.join();
    for (int fd : ofds) ::close(fd);
    for (int k = 0; k < nShards; ++k)
      if (shardResult[(size_t)k] >= 0) result = shardResult[(size_t)k];   // (of the last part that held examples)
    if (writeFailed) {
      std::cerr << "write to " << conf.output << " failed\n";
      return 1;
    }
    if (conf.timing) {
      const double wall = clock.ms();
      double analyzeMs = 0;
      for (double v : analyzeMsDev) analyzeMs = std::max(analyzeMs, v);
      std::cerr << "devices=" << nDev << " sentences=" << sentences << " gpu_ms=" << gpuUs / 1000.0 << " wall_ms=" << wall
                << " read_ms=" << (scanUs + prepUs) / 1000.0 << " analyze_ms=" << analyzeMs << " format_ms=" << formatUs / 1000.0
                << " write_ms=" << writeUs / 1000.0 << " format_count_gpu_ms=" << fmtCountUs / 1000.0 << " format_write_gpu_ms=" << fmtWriteUs / 1000.0
                << " threads=" << conf.threads << " pipeline=1 sharded=1 sent_per_s="
                << (wall > 0 ? sentences / (wall / 1000.0) : 0.0) << "\n";
      uint64_t tot[4] = {0, 0, 0, 0};
      for (auto& v : analyzers)
        for (auto& a : v)
          if (a) {
            uint64_t st[4];
            a->pipelineStats(st);
            for (int q = 0; q < 3; ++q) tot[q] += st[q];
            tot[3] = st[3];
          }
      std::cerr << "batches: one_enqueue=" << tot[0] << " rerun=" << tot[1] << " sized=" << tot[2] << " device_allocations=" << tot[3] << "\n";
      std::ostringstream ln;
      ln << "exit: process_ms_before_teardown=" << processClock.ms() << "\n";
      std::cerr << ln.str();
    }
    // The output is written and closed.  Taking the process down object by object -- four contexts with ~10 GB of
    // device buffers each, 1 GB of page-locked blocks, the mapped model -- costs 0.25 s of a 1.2 s run
    // (profiles/r05d: 811 ms at this point, 1 190 ms for the parent); the operating system and the driver release all of
    // it at process exit anyway.  --clean-exit runs the destructors (leak checkers).
    if (!conf.cleanExit) {
      if (prepin.t.joinable()) prepin.t.join();
      if (cacheWriter.t.joinable()) cacheWriter.t.join();
      std::cout.flush();
      std::cerr.flush();
      std::fflush(nullptr);
      std::_Exit(result);
    }
    return result;
  }

  // batch j goes to device j mod nDev and comes back through that device's queue: popping the done queues
  // in the same rotation restores the input order without sequence numbers
  std::vector<std::unique_ptr<BoundedQueue<std::unique_ptr<Job>>>> readQ, doneQ;
  std::vector<std::unique_ptr<Semaphore>> freeAnalyzers;
  for (int d = 0; d < nDev; ++d) {
    readQ.emplace_back(new BoundedQueue<std::unique_ptr<Job>>(2));
    doneQ.emplace_back(new BoundedQueue<std::unique_ptr<Job>>(1));
    freeAnalyzers.emplace_back(new Semaphore(nAnalyzers));
  }
  BoundedQueue<std::unique_ptr<Formatted>> writeQ(2);
  Clock clock;
  double readMs = 0, formatMs = 0, gpuMs = 0;
  std::vector<double> analyzeMsDev((size_t)nDev, 0.0);

  auto readBatch = [&](Job* job) {
    auto& batch = job->batch;
    while (batch.size() < conf.batch && hasNext()) {
      Example e;
      if (conf.partialInput) {
        e.partial.reset(new PartialExample());
        e.readStatus = pexReader.readExample(in, e.partial.get());
        batch.push_back(std::move(e));
        continue;
      }
      // PlainStreamReader::readExample (stream_reader.cc:12-38): "# " lines are the comment of the next other line
      for (;;) {
        e.input.clear();
        std::getline(*in, e.input);
        if (e.input.size() > 2 && e.input[0] == '#' && e.input[1] == ' ') std::swap(e.comment, e.input);
        else break;
      }
      if (e.comment.size() > maxComment) {
        e.readStatus = Status::InvalidParameter() << "Comment size was: " << e.comment.size() << " which is more than max: " << maxComment;
      } else if (e.input.size() > maxInput) {
        e.readStatus = Status::InvalidParameter() << "Input size was: " << e.input.size() << " which is more than max: " << maxInput;
      }
      batch.push_back(std::move(e));
    }
  };

  auto analyzeJob = [&](Job* job) {
    GpuAnalyzer& analyzer = *analyzers[job->device][job->analyzer];
    if (conf.partialInput) {
      std::vector<const PartialExample*> exs;
      for (auto& e : job->batch) exs.push_back(e.readStatus.isOk() ? e.partial.get() : nullptr);
      job->batchStatus = analyzer.analyzeBatchPartial(exs, useLattice);
    } else {
      std::vector<StringPiece> pieces;
      for (auto& e : job->batch) pieces.push_back(e.readStatus.isOk() ? StringPiece(e.input) : StringPiece(""));
      job->batchStatus = analyzer.analyzeBatch(pieces, useLattice);
    }
    float ms[8];
    analyzer.lastTimings(ms);
    job->gpuMs = ms[7];
  };

  // sentences [lo, hi) of a job -> text for stdout and stderr
  auto formatRange = [&](OutputFormat* format, const Job& job, size_t lo, size_t hi, std::string* text, std::string* errors) {
    const GpuAnalyzer& analyzer = *analyzers[job.device][job.analyzer];
    for (size_t i = lo; i < hi; ++i) {
      const Example& e = job.batch[i];
      if (!e.readStatus.isOk()) {
        *errors += "failed to read an example: " + statusText(e.readStatus);
        continue;
      }
      Status st = job.batchStatus.isOk() ? analyzer.sentenceStatus(i) : job.batchStatus;
      if (!st) {
        *errors += statusText(st);
        text->append(emptyResult.data(), emptyResult.size());
        continue;
      }
      StringPiece comment = e.comment.size() < 2 ? StringPiece("") : StringPiece(e.comment.data() + 2, e.comment.size() - 2);
      if (conf.partialInput) comment = StringPiece(e.partial->comment);
      if (analyzer.textMode()) {   // the device printed the sentence; the comment line goes in front of it
        const uint32_t* heads = analyzer.batchTextHeads();   // (lattice text: the comment replaces the "# MA-SCORE" line)
        const size_t head = heads != nullptr ? heads[i] : 0;
        size_t skip = 0;
        if (!comment.empty() && (heads == nullptr || head != 0)) {
          text->append("# ");
          text->append(comment.data(), comment.size());
          *text += '\n';
          skip = head;
        }
        const StringPiece r = analyzer.sentenceText(i);
        text->append(r.data() + skip, r.size() - skip);
        continue;
      }
      st = format->format(analyzer, i, comment);
      if (!st) *errors += statusText(st);
      else {
        StringPiece r = format->result();
        text->append(r.data(), r.size());
      }
    }
  };

  std::thread reader([&]() {
    for (size_t j = 0;; ++j) {
      std::unique_ptr<Job> job(new Job());
      double t0 = clock.ms();
      readBatch(job.get());
      readMs += clock.ms() - t0;
      if (job->batch.empty()) break;
      job->device = (int)(j % (size_t)nDev);
      readQ[job->device]->push(std::move(job));
    }
    for (auto& q : readQ) q->close();
  });
  // one analysis thread per device (HIP's current device is per host thread; the library binds it per call)
  std::vector<std::thread> gpus;
  for (int d = 0; d < nDev; ++d) {
    gpus.emplace_back([&, d]() {
      std::unique_ptr<Job> job;
      int next = 0;
      int live = nAnalyzers;  // analyzers in rotation on this device
      while (readQ[d]->pop(&job)) {
        freeAnalyzers[d]->acquire();
        job->analyzer = next;
        double t0 = clock.ms();
        if (!analyzers[d][job->analyzer]) {
          Status made = makeAnalyzer(d, job->analyzer);
          if (!made) {
            // (e.g. no HBM for a second model copy) carry on with the first analyzer alone: take the second
            // token out of circulation, which also waits until the first analyzer's batch has been formatted
            analyzers[d][job->analyzer].reset();
            freeAnalyzers[d]->acquire();
            live = 1;
            job->analyzer = 0;
          }
        }
        next = (job->analyzer + 1) % live;
        analyzeJob(job.get());
        analyzeMsDev[d] += clock.ms() - t0;
        doneQ[d]->push(std::move(job));
      }
      doneQ[d]->close();
    });
  }

  double writeMs = 0;
  std::thread writer([&]() {
    std::unique_ptr<Formatted> f;
    while (writeQ.pop(&f)) {
      double t0 = clock.ms();
      for (size_t c = 0; c < f->text.size(); ++c) {
        if (!f->errors[c].empty()) std::cerr << f->errors[c];
        out->write(f->text[c].data(), (std::streamsize)f->text[c].size());
      }
      if (conf.batch <= 16) out->flush();  // interactive use: a line's answer does not wait for the next line
      writeMs += clock.ms() - t0;
    }
  });

  int result = 0;
  size_t sentences = 0;
  const size_t kChunk = 64;  // sentences a format worker takes at a time
  std::unique_ptr<Job> job;
  for (size_t j = 0; doneQ[j % (size_t)nDev]->pop(&job); ++j) {
    double t0 = clock.ms();
    const size_t n = job->batch.size();
    const size_t nChunks = (n + kChunk - 1) / kChunk;
    std::unique_ptr<Formatted> formatted(new Formatted());
    formatted->text.resize(nChunks);
    formatted->errors.resize(nChunks);
    std::vector<std::string>& text = formatted->text;
    std::vector<std::string>& errors = formatted->errors;
    const int workers = (int)std::min<size_t>((size_t)conf.threads, nChunks);
    std::atomic<size_t> nextChunk(0);
    auto work = [&](int t) {
      for (size_t c; (c = nextChunk.fetch_add(1)) < nChunks;)
        formatRange(formats[t].get(), *job, c * kChunk, std::min(n, (c + 1) * kChunk), &text[c], &errors[c]);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < workers; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
    // the reference's exit code is that of the last example read
    result = job->batch.back().readStatus.isOk() ? 0 : 1;
    sentences += n;
    gpuMs += job->gpuMs;
    freeAnalyzers[job->device]->release();
    formatMs += clock.ms() - t0;
    writeQ.push(std::move(formatted));
  }
  writeQ.close();
  reader.join();
  for (auto& t : gpus) t.join();
  writer.join();
  out->flush();
  if (conf.timing) {
    double wall = clock.ms();
    double analyzeMs = 0;  // the busiest device's analysis stage
    for (double v : analyzeMsDev) analyzeMs = std::max(analyzeMs, v);
    std::cerr << "devices=" << nDev << " sentences=" << sentences << " gpu_ms=" << gpuMs << " wall_ms=" << wall << " read_ms=" << readMs
              << " analyze_ms=" << analyzeMs << " format_ms=" << formatMs << " write_ms=" << writeMs << " threads=" << conf.threads
              << " pipeline=" << (conf.pipeline ? 1 : 0) << " sent_per_s=" << (wall > 0 ? sentences / (wall / 1000.0) : 0.0) << "\n";
  }
  return result;
}
