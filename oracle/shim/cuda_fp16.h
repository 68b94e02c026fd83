#pragma once
struct __half { unsigned short v; };
typedef __half half;
