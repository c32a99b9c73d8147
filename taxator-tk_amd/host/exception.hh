// taxator-tk_amd/host/exception.hh -- error types of the host mirror.
// Mirrors the reference's Exception hierarchy (core/src/exception.hh:29-73) without Boost: one base class carrying the
// same kind names and the key/value diagnostics the reference attaches with boost::error_info.
#pragma once
#include <stdexcept>
#include <string>

namespace taxator {

class Exception : public std::runtime_error {
 public:
  Exception(const std::string& kind, const std::string& info) : std::runtime_error(kind + ": " + info), kind_(kind) {}
  const std::string& kind() const { return kind_; }
 private:
  std::string kind_;
};

#define TAXATOR_EXCEPTION(NAME) \
  class NAME : public Exception { public: explicit NAME(const std::string& info) : Exception(#NAME, info) {} };
TAXATOR_EXCEPTION(GeneralError)
TAXATOR_EXCEPTION(FileNotFound)
TAXATOR_EXCEPTION(FileError)
TAXATOR_EXCEPTION(ParsingError)
TAXATOR_EXCEPTION(EOFError)
TAXATOR_EXCEPTION(TaxonNotFound)
TAXATOR_EXCEPTION(TaxonMappingNotFound)
TAXATOR_EXCEPTION(SequenceNotFound)
TAXATOR_EXCEPTION(SequenceRangeError)
TAXATOR_EXCEPTION(DeviceError)
#undef TAXATOR_EXCEPTION

}  // namespace taxator
