// oracle/ref_shim/data/parser.h — TEST INFRASTRUCTURE.  Stand-ins for the two dmlc-core interfaces the
// reference's src/reader/criteo_parser.h derives from / reads through (dmlc-core is an absent submodule):
// dmlc::InputSplit (chunks of text) and dmlc::data::ParserImpl<IndexType>.  Interfaces only; the parse loop
// that is compiled is the reference's own.
#ifndef ORACLE_REF_SHIM_DATA_PARSER_H_
#define ORACLE_REF_SHIM_DATA_PARSER_H_
#include <cstddef>
#include <vector>
#include "data/row_block.h"

namespace dmlc {
class InputSplit {
 public:
  struct Blob {
    void* dptr;
    size_t size;
  };
  virtual ~InputSplit() {}
  virtual void BeforeFirst() = 0;
  virtual bool NextChunk(Blob* out_chunk) = 0;
};

namespace data {
template <typename IndexType>
class ParserImpl {
 public:
  virtual ~ParserImpl() {}
  virtual void BeforeFirst() = 0;
  virtual size_t BytesRead() const = 0;
  virtual bool ParseNext(std::vector<RowBlockContainer<IndexType> >* data) = 0;
};
}  // namespace data
}  // namespace dmlc
#endif  // ORACLE_REF_SHIM_DATA_PARSER_H_
