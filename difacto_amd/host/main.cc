/**
 * main.cc — the `difacto` command line: key=value arguments, `argfile=x.conf`
 * pulls in a config file (same "k = v" / '#' comment format as the reference's
 * example/<name>.conf), `task=train learner=sgd` by default; unknown keys are
 * reported as warnings.  Reference: src/main.cc, src/common/arg_parser.h.
 */
#include <chrono>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include "difacto/learner.h"
#include "dmlc/config.h"
#include "dmlc/parameter.h"

namespace difacto {

struct DifactoParam : public dmlc::Parameter<DifactoParam> {
  std::string task;     // train (default) | predict
  std::string learner;  // sgd
  DMLC_DECLARE_PARAMETER(DifactoParam) {
    DMLC_DECLARE_FIELD(learner).set_default("sgd");
    DMLC_DECLARE_FIELD(task).set_default("train");
  }
};
DMLC_REGISTER_PARAMETER(DifactoParam);

/*! \brief collects argv tokens and config-file text, then parses them as one "k = v" stream */
class ArgParser {
 public:
  void AddArg(const char* argv) {
    data_.append(argv);
    data_.append(" ");
  }
  KWArgs GetKWArgs() {
    std::unique_ptr<dmlc::Config> conf = Parse();
    for (const auto& it : *conf) {
      if (it.first == "argfile") {
        std::ifstream in(it.second.c_str());
        CHECK(in.good()) << "failed to open " << it.second;
        std::stringstream ss;
        ss << in.rdbuf();
        data_.append(ss.str());
        data_.append(" ");
        conf = Parse();
        break;
      }
    }
    KWArgs kwargs;
    for (const auto& it : *conf)
      if (it.first != "argfile") kwargs.push_back(it);
    return kwargs;
  }

 private:
  std::unique_ptr<dmlc::Config> Parse() {
    std::stringstream ss(data_);
    return std::unique_ptr<dmlc::Config>(new dmlc::Config(ss));
  }
  std::string data_;
};

void WarnUnknownKWArgs(const DifactoParam& param, const KWArgs& remain) {
  if (remain.empty()) return;
  LOG(WARNING) << "Unrecognized keyword argument for task = " << param.task;
  for (const auto& kw : remain) LOG(WARNING) << " - " << kw.first << " = " << kw.second;
}

}  // namespace difacto

int main(int argc, char* argv[]) {
  if (argc < 2) {
    LOG(ERROR) << "usage: difacto key1=val1 key2=val2 ...";
    return 0;
  }
  using namespace difacto;
  // DIFACTO_PROFILE=1: where the PROCESS spends its wall-clock outside the worker loop (start-up and teardown are most of a
  // one-epoch job: profiles/r05e_e2e_startup.txt)
  const bool prof = getenv("DIFACTO_PROFILE") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  ArgParser parser;
  for (int i = 1; i < argc; ++i) parser.AddArg(argv[i]);
  DifactoParam param;
  auto kwargs_remain = param.InitAllowUnknown(parser.GetKWArgs());
  if (param.task == "train") {
    Learner* learner = Learner::Create(param.learner);
    WarnUnknownKWArgs(param, learner->Init(kwargs_remain));
    const double t1 = now();
    learner->Run();
    const double t2 = now();
    delete learner;
    if (prof)
      LOG(INFO) << "process: main() to the learner initialised (HIP runtime, device context, model table) " << t1 - t0 << " s, Run "
                << t2 - t1 << " s, learner destroyed in " << now() - t2 << " s";
  } else if (param.task == "predict") {
    // the reference stops at a TODO here (main.cc:61-62); its SGDLearnerParam already names model_in as the model of
    // "a prediction task" (sgd_param.h:24-28).  The learner runs one forward pass over the data and writes pred_out.
    Learner* learner = Learner::Create(param.learner);
    kwargs_remain.push_back(std::make_pair("task", "predict"));
    WarnUnknownKWArgs(param, learner->Init(kwargs_remain));
    learner->Run();
    delete learner;
  } else {
    LOG(FATAL) << "unknown task: " << param.task << " (this build provides train and predict)";
  }
  return 0;
}
