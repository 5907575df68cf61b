// TEST INFRASTRUCTURE: stand-in for the header jrl-cmakemodules generates at configure time (the submodule is
// empty in /root/reference and cmake is not run): export / version macros only.
#pragma once
#define ALIGATOR_VERSION "0.0.0-ref-shim"
#define ALIGATOR_MAJOR_VERSION 0
#define ALIGATOR_MINOR_VERSION 0
#define ALIGATOR_PATCH_VERSION 0
#define ALIGATOR_DLLAPI
#define ALIGATOR_DLLIMPORT
#define ALIGATOR_DLLEXPORT
#define ALIGATOR_DLLLOCAL
#define ALIGATOR_PRAGMA(x) _Pragma(#x)
// (the generated header also carries these two: empty outside Windows)
#define ALIGATOR_EXPLICIT_INSTANTIATION_DECLARATION_DLLAPI
#define ALIGATOR_EXPLICIT_INSTANTIATION_DEFINITION_DLLAPI
