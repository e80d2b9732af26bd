// tests/mock_ref/utility.h -- TEST-ONLY: the adapter includes the reference's utility.h but uses nothing of it.
#ifndef MOCK_REF_UTILITY_H__
#define MOCK_REF_UTILITY_H__
#endif
