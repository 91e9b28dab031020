// TEST INFRASTRUCTURE — stand-in for <message_filters/subscriber.h> (included by feature_tracker_node.cpp, nothing of it is used)
#ifndef VINS_REF_FE_MESSAGE_FILTERS_SUBSCRIBER_H
#define VINS_REF_FE_MESSAGE_FILTERS_SUBSCRIBER_H
#endif
