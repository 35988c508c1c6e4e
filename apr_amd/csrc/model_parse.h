// model_parse.h -- weight ingest shared by the engine: ONNX (ai/model.py:143-152 export)
// or the flat APRW blob (apr_amd/weights.py) -> BN-folded conv chain + classifier.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace apr {

constexpr int kMaxConvWidth = 4096;   // channels of one conv layer / features of the classifier (the reference's widest is 1024)
constexpr int kMaxClasses = 64;       // = APR_MAX_CLASSES (include/apr.h)

struct ConvSpec {
    int cin = 0, cout = 0, stride = 1;
    std::vector<float> w;   // [cout][cin][3][3], BatchNorm folded in
    std::vector<float> b;   // [cout]
};

struct ParsedModel {
    std::vector<ConvSpec> convs;
    int n_classes = 0, features = 0;
    std::vector<float> fc_w;   // [n_classes][features]
    std::vector<float> fc_b;   // [n_classes]
};

// Returns true on success; on failure `err` says what was wrong.
bool parse_model(const void *bytes, size_t n, ParsedModel &out, std::string &err);

}  // namespace apr
