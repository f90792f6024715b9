// TEST INFRASTRUCTURE -- placeholder, filled in by the Lineq restatement.
#ifndef XPOLY_ORACLE_LINEQ_H
#define XPOLY_ORACLE_LINEQ_H
#endif
