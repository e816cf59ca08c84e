// cupoch/utility/console.h -- the slice of the reference's logging interface the registration path uses
// (src/cupoch/utility/console.h:30-71: VerbosityLevel, Set/GetVerbosityLevel).  At Debug verbosity
// registration::RegistrationICP reports every iteration like the reference does (registration.cu:155-156).
#pragma once

namespace cupoch {
namespace utility {

enum class VerbosityLevel { Trace = 0, Debug = 1, Info = 2, Warning = 3, Error = 4, Critical = 5, Off = 6 };

void SetVerbosityLevel(VerbosityLevel level);
VerbosityLevel GetVerbosityLevel();

}  // namespace utility
}  // namespace cupoch
