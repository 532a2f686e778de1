// Wrapper translation unit: compiles the REFERENCE's own CStereoMatching.cpp where it lies
// (/root/reference is never copied).  The only trick is access: the stage functions are private
// members (CStereoMatching.h:49-70); the class header is pulled in with `private` spelled `public`
// AFTER the standard / OpenCV / Armadillo headers have been included normally.
#define __declspec(x)
#define _Longlong long long
#include "SharedInclude.h"   // reference/reconstruction: std + vendored OpenCV 2.4.5 headers + Armadillo 4.200
#define private public
#include "CStereoMatching.h"
#undef private
#include "CStereoMatching.cpp"
