/**
 * libsvm_reader.h — the libsvm-text instance of BatchReader (batch_reader.h), kept under the name the
 * host tests and the literal worker loop use.  "label idx:val ..." lines, feature indices taken as
 * they are (no 1-based shift), as the reference does.
 */
#ifndef DIFACTO_HOST_LIBSVM_READER_H_
#define DIFACTO_HOST_LIBSVM_READER_H_
#include <string>
#include "./batch_reader.h"

namespace difacto {

class LibsvmBatchReader : public BatchReader {
 public:
  LibsvmBatchReader(const std::string& uri, unsigned part_index, unsigned num_parts, unsigned batch_size,
                    unsigned shuffle_buf_size = 0, float neg_sampling = 1.0f)
      : BatchReader(uri, "libsvm", part_index, num_parts, batch_size, shuffle_buf_size, neg_sampling) {}
};

}  // namespace difacto
#endif  // DIFACTO_HOST_LIBSVM_READER_H_
