// Status: error reporting of the host layer.  Same kinds and the same
// "no exceptions on the hot path" discipline as the reference's
// util/status.hpp (every stage returns Status; JPP_RETURN_IF_ERROR).
#ifndef JUMANPP_AMD_HOST_STATUS_H
#define JUMANPP_AMD_HOST_STATUS_H

#include <ostream>
#include <sstream>
#include <string>

namespace jumanpp_amd {

enum class StatusCode : int {
  Ok = 0,
  InvalidParameter = 1,
  InvalidState = 2,
  NotImplemented = 3,
  EndOfIteration = 4,
  NoDevice = 5,
  OutOfMemory = 6,
};

class Status {
  StatusCode code_ = StatusCode::Ok;
  std::string message_;

 public:
  Status() = default;
  Status(StatusCode c, std::string msg) : code_(c), message_(std::move(msg)) {}
  static Status Ok() { return Status(); }
  static Status InvalidParameter(std::string m = std::string()) { return Status(StatusCode::InvalidParameter, std::move(m)); }
  static Status InvalidState(std::string m = std::string()) { return Status(StatusCode::InvalidState, std::move(m)); }
  static Status NotImplemented(std::string m = std::string()) { return Status(StatusCode::NotImplemented, std::move(m)); }
  bool isOk() const { return code_ == StatusCode::Ok; }
  explicit operator bool() const { return isOk(); }
  StatusCode code() const { return code_; }
  const std::string& message() const { return message_; }

  template <typename T>
  Status& operator<<(const T& v) {
    std::ostringstream s;
    s << v;
    message_ += s.str();
    return *this;
  }
};

inline const char* statusName(StatusCode c) {
  switch (c) {
    case StatusCode::Ok: return "Ok";
    case StatusCode::InvalidParameter: return "InvalidParameter";
    case StatusCode::InvalidState: return "InvalidState";
    case StatusCode::NotImplemented: return "NotImplemented";
    case StatusCode::EndOfIteration: return "EndOfIteration";
    case StatusCode::NoDevice: return "NoDevice";
    case StatusCode::OutOfMemory: return "OutOfMemory";
  }
  return "?";
}

inline std::ostream& operator<<(std::ostream& o, const Status& s) {
  return o << statusName(s.code()) << ": " << s.message();
}

#define JPPA_RETURN_IF_ERROR(expr)                    \
  do {                                                \
    ::jumanpp_amd::Status _jppa_status = (expr);      \
    if (!_jppa_status.isOk()) return _jppa_status;    \
  } while (0)

// non-owning (pointer, length) view; the host layer is C++14, so no std::string_view
struct StringPiece {
  const char* ptr = nullptr;
  size_t len = 0;
  StringPiece() = default;
  StringPiece(const char* p, size_t n) : ptr(p), len(n) {}
  StringPiece(const std::string& s) : ptr(s.data()), len(s.size()) {}  // NOLINT: implicit like the reference's
  StringPiece(const char* z) : ptr(z), len(z ? std::char_traits<char>::length(z) : 0) {}  // NOLINT
  const char* data() const { return ptr; }
  size_t size() const { return len; }
  bool empty() const { return len == 0; }
  char operator[](size_t i) const { return ptr[i]; }
  std::string str() const { return std::string(ptr, len); }
};

inline std::ostream& operator<<(std::ostream& o, const StringPiece& s) {
  return o.write(s.ptr, (std::streamsize)s.len);
}

}  // namespace jumanpp_amd

#endif  // JUMANPP_AMD_HOST_STATUS_H
