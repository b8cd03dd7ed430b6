// Common/Errata.h -- error convention of the reference (OpenCLRayTracer/Include/Common/Errata.h:36-137),
// re-implemented: enum Result {Success, Error}, class Errata (message, file, function, line, exception),
// CLInterfaceException, FILL_ERRATA, operator<<.  Same names and meaning so that code written against the
// reference's managers compiles unchanged.
#ifndef RTMI_HOST_ERRATA_H
#define RTMI_HOST_ERRATA_H
#include <exception>
#include <ostream>
#include <sstream>
#include <string>

#define FILL_ERRATA(err, errorMessage)                                                                         \
    {                                                                                                          \
        std::stringstream msg__;                                                                               \
        msg__ << errorMessage;                                                                                 \
        err = CLRayTracer::Common::Errata(msg__.str(), std::string(__FILE__), std::string(__func__), __LINE__); \
    }

namespace CLRayTracer { namespace Common {

enum Result { Success = 0, Error = 1 };

class Errata {
public:
    Errata() : _line(0), _exception(nullptr) {}
    Errata(const std::string& errorMessage, const std::string& file, const std::string& function, int line,
           const std::exception* ex = nullptr)
        : _message(errorMessage), _file(file), _function(function), _line(line), _exception(ex) {}
    std::string message() const { return _message; }
    std::string file() const { return _file; }
    std::string function() const { return _function; }
    int line() const { return _line; }
    const std::exception* exception() const { return _exception; }
protected:
    std::string _message, _file, _function;
    int _line;
    const std::exception* _exception;
    friend std::ostream& operator<<(std::ostream& o, const Errata& err);
};

class CLInterfaceException : public std::exception, public Errata {
public:
    CLInterfaceException(const std::string& errorMessage, const std::string& file, const std::string& function, int line)
        : Errata(errorMessage, file, function, line) {}
    explicit CLInterfaceException(const Errata& err) : Errata(err) {}
    const char* what() const noexcept override { return _message.c_str(); }
};

inline std::ostream& operator<<(std::ostream& o, const Errata& err)
{
    o << err._file << '(' << err._line << ')' << ' ' << err._function << ' ' << err._message << ' '
      << (err._exception ? err._exception->what() : "") << std::endl;
    return o;
}

}}  // namespace
#endif
